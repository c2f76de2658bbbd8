#!/bin/bash
# conv_pk stores the activated output's bf16 plane (d_conv_0 + LeakyReLU -> d_conv_1): bf16 parity, in-box A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04ac; mkdir -p $O
timeout 1200 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/pytest_bf16.txt 2>&1
tail -3 $O/pytest_bf16.txt
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_pk_plane.txt; }
for r in 1 2 3; do
  for e in "EG_X=1" "EG_NO_PK_PLANE=1"; do run "$e" "--dtype bf16"; done
done
