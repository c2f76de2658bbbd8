#!/bin/bash
# split-count A/Bs (experiments build): plane-fed filter gradient 768 / 512 / 384 / 256 blocks (bf16), packed thin filter gradient 512 / 256 / 1024 blocks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04u; mkdir -p $O
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_splits.txt; }
for r in 1 2; do
  for t in 768 512 384 256; do run "EG_WGRAD_BF16P_BLOCKS=$t" "--dtype bf16"; done
  for t in 512 256 1024; do run "EG_WGRAD_PK_BLOCKS=$t" "--dtype f32x3"; done
done
