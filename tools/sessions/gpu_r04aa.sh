#!/bin/bash
# overlap analysis: kernel traces with lanes on / off, both modes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04aa; mkdir -p $O
for m in f32x3 bf16; do
rocprofv3 --kernel-trace -d /tmp/pc_$m -o trace -- python bench.py --dtype $m --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $O/conc_$m.json 2> $O/conc_$m.err
EG_LANES=0 rocprofv3 --kernel-trace -d /tmp/ps_$m -o trace -- python bench.py --dtype $m --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $O/ser_$m.json 2> $O/ser_$m.err
W=$(python -c "import json;d=json.loads(open('$O/conc_$m.json').read().strip().splitlines()[-1]);print(d['ms_per_step']*d['steps'])")
python tools/trace_overlap.py $(find /tmp/pc_$m -name '*.db' | head -1) $(find /tmp/ps_$m -name '*.db' | head -1) 40 $W > $O/overlap_$m.txt 2>&1
done
head -8 $O/overlap_f32x3.txt
