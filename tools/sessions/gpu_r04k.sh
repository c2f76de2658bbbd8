#!/bin/bash
# round 4, step k: the whole GPU suite at this HEAD, then both bench lines
cd /root/repo; mkdir -p gpurun_out/r04k; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -45 > gpurun_out/r04k/pytest_gpu.txt
tail -8 gpurun_out/r04k/pytest_gpu.txt | cut -c1-200
python bench.py --no-cpu-baseline > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04k/bench_bf16.json 2>> gpurun_out/r04k/bench.err
cut -c1-220 gpurun_out/r04k/bench.json gpurun_out/r04k/bench_bf16.json
