#!/bin/bash
# Serialised (EG_LANES=0) rocprofv3 kernel trace of the bench for one library build -> per-kernel CSV.
# usage (on the GPU box): bash tools/gpu_trace.sh <out.csv> [lib.so] [extra bench args...]
OUT=$1; LIB=${2:-edgegan_amd/libedgegan_hip.so}; shift; shift
export TMPDIR=/tmp
D=/tmp/prof_$$
EG_LIB_PATH=$PWD/$LIB EG_LANES=0 rocprofv3 --kernel-trace --stats -d $D -o trace -- python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt "$@" > ${OUT%.csv}.bench.json 2> ${OUT%.csv}.err
DB=$(find $D -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB $OUT 13
rm -rf $D
