#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_NO_DEFER_SLOPES=1 EG_LANES=15 5 2>&1 | tee gpurun_out/r03x/ab_defer_slopes.txt
unset EG_LIB_PATH
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_tight_gpu.py tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/r03x/pytest_defer_slopes.txt
