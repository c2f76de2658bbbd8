#!/bin/bash
# at the round's HEAD: the driver's own bench command, smoke, the whole GPU suite
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06zz
( time timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06zz/bench_driver_cmd.json 2> gpurun_out/r06zz/bench_driver_cmd.err
python -c "import json;d=json.load(open('gpurun_out/r06zz/bench_driver_cmd.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['bf16_mode']['value'], d['config4_256x512_b16_bf16']['value'], d['config1_infer_b64']['value'], d['bench_seconds'])"
tail -3 gpurun_out/r06zz/bench_driver_cmd.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06zz/smoke.txt 2>&1; tail -1 gpurun_out/r06zz/smoke.txt | cut -c1-120
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06zz/pytest_gpu.txt 2>&1
grep -E "passed|failed" gpurun_out/r06zz/pytest_gpu.txt
