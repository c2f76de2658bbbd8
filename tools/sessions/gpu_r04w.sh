#!/bin/bash
# conv_scatter on limb MFMAs: op parity (f32x3, bf16), layer table, in-box A/B (EG_SCATTER_F32=1 = the fp32-MFMA form)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04w; mkdir -p $O
timeout 900 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_x3_gpu.py::test_x3_network_forwards -m gpu -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python tools/layer_table.py $O/layers.csv > $O/layers.txt 2>&1
grep "scatter" $O/layers.csv | cut -c1-200
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_scatter.txt; }
for r in 1 2; do
  for e in "EG_X=1" "EG_SCATTER_F32=1"; do run "$e" "--dtype f32x3"; run "$e" "--dtype bf16"; done
done
