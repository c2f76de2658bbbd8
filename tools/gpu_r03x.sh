#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
DTYPE=f32x3 SHAPES='cls Conv_1 u1,D.conv0,cls h0' timeout 300 python tools/gpu_microbench.py 2>&1 | grep -i "smallc\|scatter\|flatm" | tee gpurun_out/r03x/smallc_batched_gather.txt
unset EG_LIB_PATH
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee -a gpurun_out/r03x/smallc_batched_gather.txt
