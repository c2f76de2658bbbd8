#!/bin/bash
# f32x3: weight of the slab-reduction term in the split-K cost model (1 = as calibrated in round 3)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04y; mkdir -p $O
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_slab_cost.txt; }
for r in 1 2; do
  for t in 1.0 2.0 4.0 0.5; do run "EG_X3_SLAB_COST=$t" "--dtype f32x3"; done
done
