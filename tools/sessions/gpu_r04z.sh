#!/bin/bash
# round 4: the final profile sets, f32x3 (headline) and bf16
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_round_profile.sh r04z > gpurun_out/r04z_profile.log 2>&1
bash tools/gpu_round_profile_bf16.sh r04z > gpurun_out/r04z_profile_bf16.log 2>&1
tail -2 gpurun_out/r04z_profile.log; cut -c1-200 gpurun_out/r04z/bench.json gpurun_out/r04z/bench_bf16.json
