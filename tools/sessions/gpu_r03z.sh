#!/bin/bash
# final set of the round: full GPU suite, then the round profile (PMC passes, bench lines, serialised trace, layer table)
cd /root/repo; mkdir -p gpurun_out/r03z
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r03z/pytest_gpu.txt
cat gpurun_out/r03z/pytest_gpu.txt
bash tools/gpu_round_profile.sh r03z > gpurun_out/r03z/profile.log 2>&1
tail -3 gpurun_out/r03z/profile.log
cat gpurun_out/r03z/bench.json | tail -1 | cut -c1-400
