#!/bin/bash
# round 4, step s: thin layers, second pass (wgrad_pk + generalised conv_pk): op parity, layer table, in-box A/B against the
# round's first version (EG_PK_FWD_ONLY=1 EG_NO_WGRAD_PK=1 in the experiments build)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04s; mkdir -p $O
timeout 900 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle -m gpu -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
python tools/layer_table.py $O/layers.csv > $O/layers.txt 2>&1
grep "conv_pk\|smallc" $O/layers.csv | cut -c1-200
for r in 1 2; do
for env in "EG_X=1" "EG_PK_FWD_ONLY=1 EG_NO_WGRAD_PK=1"; do
  env $env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32x3 [$env]', d['value'])" | tee -a $O/ab_thin2.txt
  env $env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 [$env]', d['value'])" | tee -a $O/ab_thin2.txt
done; done
