#!/bin/bash
# deferred (batched) vs immediate second-stage reductions, both modes (the slabs of an immediate reduction are still in the Infinity Cache)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04af; mkdir -p $O
run() { env $1 python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_defer.txt; }
for r in 1 2; do for e in "EG_X=1" "EG_DEFER_REDUCE=0"; do run "$e" "--dtype f32x3"; run "$e" "--dtype bf16"; done; done
