#!/bin/bash
# Quick serialised kernel trace + per-layer table of the current build.  usage (GPU box): bash tools/gpu_quick_profile.sh <tag>
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/layer_table.py $OUT/layers.csv 2> $OUT/layers.err
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/kernel_stats.bench.json 2> $OUT/prof.err
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB $OUT/kernel_stats.csv 13; fi
python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/bench.json 2>> $OUT/prof.err
