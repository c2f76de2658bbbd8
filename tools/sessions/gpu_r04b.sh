#!/bin/bash
# round 4, step b: the new host-side tests (context recovery, pipelined pair, self-launching bench) + a fresh bf16 profile
# set (serialised kernel trace, per-layer table, PMC utilisation) -- VERDICT r3: no bf16 trace existed for round 3
cd /root/repo; mkdir -p gpurun_out/r04b; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_gpu.py::test_context_recovers_from_an_error_in_the_middle_of_a_backward_sweep tests/test_step_gpu.py::test_pipelined_feed_and_loss_readback_equal_the_synchronous_calls tests/test_dp_gpu.py::test_bench_launches_its_own_ranks -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04b/pytest_new.txt
cat gpurun_out/r04b/pytest_new.txt
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -o trace -- python bench.py --dtype bf16 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > gpurun_out/r04b/kernel_stats_bf16.bench.json 2> gpurun_out/r04b/prof.err
DB=$(find /tmp/prof_bf -name '*.db' | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r04b/kernel_stats_bf16.csv 13
python tools/layer_table.py gpurun_out/r04b/layers_bf16.csv --dtype bf16 2> gpurun_out/r04b/layers.err
EG_LANES=0 bash tools/gpu_pmc.sh gpurun_out/r04b/pmc "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" -- python bench.py --dtype bf16 --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-alt > gpurun_out/r04b/pmc.log 2>&1
python tools/pmc_derive.py gpurun_out/r04b/pmc/summary.csv gpurun_out/r04b/pmc_utilisation_bf16.csv
head -30 gpurun_out/r04b/kernel_stats_bf16.csv | cut -c1-200
