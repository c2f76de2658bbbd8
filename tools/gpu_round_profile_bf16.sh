#!/bin/bash
# The bf16 mode's per-round artefacts (none existed for round 3): PMC passes (MFMA utilisation, HBM traffic, LDS),
# serialised rocprofv3 kernel trace, per-layer table, bench line.   usage (on the GPU box): bash tools/gpu_round_profile_bf16.sh <tag>
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
EG_LANES=0 bash tools/gpu_pmc.sh $OUT/pmc_bf16 "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" -- python bench.py --no-side-legs --dtype bf16 --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/pmc_bf16.log 2>&1
python tools/pmc_derive.py $OUT/pmc_bf16/summary.csv $OUT/pmc_utilisation_bf16.csv
python bench.py --no-side-legs --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python tools/layer_table.py $OUT/layers_bf16.csv --dtype bf16 2> $OUT/layers_bf16.err
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf_$TAG -o trace -- python bench.py --no-side-legs --dtype bf16 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/kernel_stats_bf16.bench.json 2> $OUT/prof_bf16.err
DB=$(find /tmp/prof_bf_$TAG -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB $OUT/kernel_stats_bf16.csv 13; fi
