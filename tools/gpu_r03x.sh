#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
timeout 900 python tools/gpu_ab_env.py EG_NO_SLAB_IN_TRAIN=1 EG_LANES=15 3 2>&1 | tee gpurun_out/r03x/ab_slab_in_train.txt
timeout 600 python tools/gpu_ab_env.py EG_NO_SLAB_IN_TRAIN=1 EG_LANES=15 2 --dtype bf16 2>&1 | tee gpurun_out/r03x/ab_slab_in_train_bf16.txt
timeout 600 python tools/gpu_ab_env.py EG_NO_SLAB_IN_TRAIN=1 EG_LANES=15 2 --height 256 --width 512 --batch 16 2>&1 | tee gpurun_out/r03x/ab_slab_in_train_256.txt
unset EG_LIB_PATH
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_golden_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/r03x/pytest_step_golden.txt
