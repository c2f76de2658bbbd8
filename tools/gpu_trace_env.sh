#!/bin/bash
# like tools/gpu_trace.sh, with extra environment assignments: bash tools/gpu_trace_env.sh <out.csv> <lib.so> "A=1 B=2" [bench args...]
OUT=$1; LIB=$2; ENVS=$3; shift; shift; shift
export TMPDIR=/tmp
D=/tmp/prof_$$
env $ENVS EG_LIB_PATH=$PWD/$LIB EG_LANES=0 rocprofv3 --kernel-trace --stats -d $D -o trace -- python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt "$@" > ${OUT%.csv}.bench.json 2> ${OUT%.csv}.err
DB=$(find $D -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT 13
rm -rf $D
