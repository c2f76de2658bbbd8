#!/bin/bash
# round 4, step l: split-K choice of the plane-fed kernel, 288-row halo, the batch-16 bf16 parity tests, the sharded CPU baseline
cd /root/repo; mkdir -p gpurun_out/r04l; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_bf16_gpu.py -x -q -p no:cacheprovider -s --durations=12 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r04l/pytest_bf16.txt
grep -E "passed|failed|bf16 B=16|s call" gpurun_out/r04l/pytest_bf16.txt | cut -c1-200
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04l/bench_bf16.json 2> gpurun_out/r04l/bench.err
cut -c1-200 gpurun_out/r04l/bench_bf16.json
python tools/layer_table.py gpurun_out/r04l/layers_bf16.csv --dtype bf16 2> gpurun_out/r04l/layers.err
python bench.py > gpurun_out/r04l/bench.json 2>> gpurun_out/r04l/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r04l/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], json.dumps(d['cpu_baseline'])[:1500])"
