#!/bin/bash
# round 4, step m: packed-K thin forward kernel (k_thin_pk.hip): op parity in both modes, in-box A/B, layer table; batch-16 bf16 tests
cd /root/repo; mkdir -p gpurun_out/r04m; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_bf16_gpu.py::test_bf16_substep_at_batch16_against_the_emulating_oracle tests/test_x3_gpu.py::test_x3_network_forwards tests/test_x3_gpu.py::test_x3_substeps_14_classes -x -q -p no:cacheprovider -s 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r04m/pytest.txt
grep -E "passed|failed|bf16 B=16|FAIL" gpurun_out/r04m/pytest.txt | cut -c1-220
for env in "" "EG_NO_PK=1"; do
  env $env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> gpurun_out/r04m/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32x3 [$env]', d['value'])" | tee -a gpurun_out/r04m/ab_pk.txt
  env $env EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 --steps 20 2>> gpurun_out/r04m/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 [$env]', d['value'])" | tee -a gpurun_out/r04m/ab_pk.txt
done
python tools/layer_table.py gpurun_out/r04m/layers.csv 2> gpurun_out/r04m/layers.err
grep conv_pk gpurun_out/r04m/layers.csv | cut -c1-170
