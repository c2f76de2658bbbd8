#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_NO_RESIZE_X4=1 EG_LANES=15 3 2>&1 | tee gpurun_out/r03x/ab_resize_x4.txt
