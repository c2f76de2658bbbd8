#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
