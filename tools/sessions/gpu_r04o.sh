#!/bin/bash
# round 4, step o: narrow (128 x 64) un-split tiles for small layers (A/B), tightened full-batch tolerances
cd /root/repo; mkdir -p gpurun_out/r04o; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_bf16_gpu.py::test_bf16_substeps_14_classes tests/test_tight_gpu.py::test_full_batch_fp32_substep_against_fp64_oracle tests/test_x3_gpu.py::test_x3_full_batch_substep_against_fp64_oracle -x -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r04o/pytest.txt
tail -3 gpurun_out/r04o/pytest.txt | cut -c1-200
for env in "" "EG_BF16P_NO_NARROW=1"; do
  env $env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 --steps 20 2>> gpurun_out/r04o/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 [$env]', d['value'])" | tee -a gpurun_out/r04o/ab_narrow.txt
done
env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 --steps 20 2>> gpurun_out/r04o/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 [again]', d['value'])" | tee -a gpurun_out/r04o/ab_narrow.txt
python tools/layer_table.py gpurun_out/r04o/layers_bf16.csv --dtype bf16 2> gpurun_out/r04o/layers.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04o/bench_bf16.json 2>> gpurun_out/r04o/bench.err
