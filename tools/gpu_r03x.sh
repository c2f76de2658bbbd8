#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -3 | tee gpurun_out/r03x/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
python bench.py --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | cut -c1-300
