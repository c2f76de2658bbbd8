#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_DUP_PRELU_FWD=0 EG_DUP_PRELU_FWD=2 4 2>&1 | tee gpurun_out/r03x/ab_dup_prelu_fwd.txt
timeout 900 python tools/gpu_ab_env.py EG_DUP_PRELU_FWD=0 EG_DUP_PRELU_FWD=6 3 2>&1 | tee -a gpurun_out/r03x/ab_dup_prelu_fwd.txt
