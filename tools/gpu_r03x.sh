#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_NO_X3H_PAIR=1 EG_LANES=15 4 2>&1 | tee gpurun_out/r03x/ab_x3h_pair.txt
timeout 600 python tools/gpu_ab_env.py EG_NO_X3H_PAIR=1 EG_LANES=15 2 --height 256 --width 512 --batch 16 2>&1 | tee gpurun_out/r03x/ab_x3h_pair_256.txt
