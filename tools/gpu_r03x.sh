#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_X3_TILE1_COST=2.0 EG_X3_TILE1_COST=1.7 6 2>&1 | tee -a gpurun_out/r03x/ab_tile1_cost.txt
timeout 900 python tools/gpu_ab_env.py EG_X3_TILE1_COST=2.0 EG_X3_TILE1_COST=1.8 6 2>&1 | tee -a gpurun_out/r03x/ab_tile1_cost.txt
