#!/bin/bash
# rocprofv3 --pmc passes (one per counter group, nothing else traced) over a command; per-kernel averages as CSV.
# usage: bash tools/gpu_pmc.sh <out_dir> "<counter group 1>" "<counter group 2>" ... -- <command...>
OUT=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for g in "${GROUPS_[@]}"; do
  d=/tmp/pmc_$$_$i
  rocprofv3 --pmc $g -d $d -o p --output-format csv -- "$@" > $OUT/run_$i.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/group_$i.csv
  i=$((i+1))
done
python tools/pmc_table.py $OUT > $OUT/summary.csv
cat $OUT/summary.csv
