#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/r03x/pytest_gpu.txt
python tools/layer_table.py gpurun_out/r03x/layers.csv 2> /dev/null
