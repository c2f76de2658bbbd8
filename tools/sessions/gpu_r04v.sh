#!/bin/bash
# split-K target of the plane-fed convolution kernel (bf16): 512 / 384 / 256 / 768 blocks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04v2; mkdir -p $O
run() { env $1 EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py $2 --no-cpu-baseline --no-alt --profile-steps 0 --steps 20 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 [$1]', d['value'])" | tee -a $O/ab_ksplit.txt; }
for r in 1 2; do
  for t in 256 192 128 320; do run "EG_BF16P_KS_BLOCKS=$t" "--dtype bf16"; done
done
