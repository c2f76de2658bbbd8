#!/bin/bash
# in-box A/B of three builds in the bf16 mode (why did the bf16 side leg read 2,818 on one box after 3,258 / 3,274 on others?)
cd /root/repo
mkdir -p gpurun_out/r06e
python tools/gpu_ab.py edgegan_amd/libedgegan_hip.so edgegan_amd/libedgegan_hip_prev.so 3 --dtype bf16 --no-alt --no-side-legs --no-live-traffic > gpurun_out/r06e/ab_bf16_head_vs_prev.txt 2>&1
cat gpurun_out/r06e/ab_bf16_head_vs_prev.txt
python tools/gpu_ab.py edgegan_amd/libedgegan_hip.so edgegan_amd/libedgegan_hip_prev2.so 3 --dtype bf16 --no-alt --no-side-legs --no-live-traffic > gpurun_out/r06e/ab_bf16_head_vs_gp.txt 2>&1
cat gpurun_out/r06e/ab_bf16_head_vs_gp.txt
python tools/gpu_ab.py edgegan_amd/libedgegan_hip.so edgegan_amd/libedgegan_hip_prev2.so 3 --no-alt --no-side-legs --no-live-traffic > gpurun_out/r06e/ab_f32x3_head_vs_gp.txt 2>&1
cat gpurun_out/r06e/ab_f32x3_head_vs_gp.txt
