#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_PRIO_LANE=-1 EG_PRIO_LANE=3 4 2>&1 | tee gpurun_out/r03x/ab_prio_lane.txt
timeout 900 python tools/gpu_ab_env.py EG_PRIO_LANE=-1 EG_PRIO_LANE=1 3 2>&1 | tee -a gpurun_out/r03x/ab_prio_lane.txt
