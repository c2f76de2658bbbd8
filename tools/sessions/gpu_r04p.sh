#!/bin/bash
# round 4, step p: the round's profile sets, f32x3 (headline) and bf16
cd /root/repo
bash tools/gpu_round_profile.sh r04p > gpurun_out/r04p_profile.log 2>&1
bash tools/gpu_round_profile_bf16.sh r04p > gpurun_out/r04p_profile_bf16.log 2>&1
tail -2 gpurun_out/r04p_profile.log; cut -c1-200 gpurun_out/r04p/bench.json gpurun_out/r04p/bench_bf16.json
