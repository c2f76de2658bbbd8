#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03w
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_WGRAD_ROWHALO=0 EG_WGRAD_ROWHALO=1 3 2>&1 | tee gpurun_out/r03w/ab_rowhalo_s1.txt
timeout 900 python tools/gpu_ab_env.py EG_WGRAD_ROWHALO=1 EG_WGRAD_ROWHALO=3 3 2>&1 | tee gpurun_out/r03w/ab_rowhalo_s2.txt
timeout 600 python tools/gpu_ab_env.py EG_WGRAD_ROWHALO=0 EG_WGRAD_ROWHALO=3 2 --height 256 --width 512 --batch 16 2>&1 | tee gpurun_out/r03w/ab_rowhalo_256.txt
unset EG_LIB_PATH
timeout 1500 python -m pytest tests/ -x -q -m gpu --deselect tests/test_x3_gpu.py --deselect tests/test_tight_gpu.py 2>&1 | tail -5 | tee gpurun_out/r03w/pytest_gpu_rest.txt
