#!/bin/bash
# x3q (two blocks per CU, 128 x 128 tiles, 16-channel chunks): op parity, microbench vs the halo kernel, in-box A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04ad2; mkdir -p $O
timeout 900 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt | cut -c1-250
for e in "EG_X=1" "EG_NO_X3Q=1"; do
  echo "== $e" | tee -a $O/microbench.txt
  env $e EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so DTYPE=f32x3 SHAPES="cls Conv_2,E.res 3x3 REFLECT 128,D.conv3,D.conv4 patch" python tools/gpu_microbench.py 2>/dev/null | grep -v "^shape" | grep "x3h\|x3q" | tee -a $O/microbench.txt | cut -c1-110
done
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_x3q.txt; }
for r in 1 2; do for e in "EG_X=1" "EG_NO_X3Q=1"; do run "$e" "--dtype f32x3"; done; done
