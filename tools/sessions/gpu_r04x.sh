#!/bin/bash
# serialised kernel traces of both modes after the thin-layer / split work (where the time is now)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04x; mkdir -p $O
for m in f32x3 bf16; do
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o trace -- python bench.py --dtype $m --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $O/kernel_stats_$m.bench.json 2> $O/prof_$m.err
DB=$(find /tmp/prof_$m -name '*.db' | head -1)
python tools/rocprof_summary.py $DB $O/kernel_stats_$m.csv 13
done
