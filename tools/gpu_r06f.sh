#!/bin/bash
# round 6: the default bench with its side legs as fresh processes; inference; the step / dp / golden tests after removing gen_head
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06f
( time timeout 400 python bench.py ) > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
python -c "import json;d=json.load(open('gpurun_out/r06f/bench.json'));print(d['value'], d['bf16_mode']['value'], d['config4_256x512_b16_bf16']['value'], d['config1_infer_b64']['value'], d['bench_seconds'], d['cpu_baseline']['value'])"
tail -4 gpurun_out/r06f/bench.err
timeout 300 python tools/gpu_infer_fusion_check.py > gpurun_out/r06f/infer_fusion_check.txt 2>&1
tail -6 gpurun_out/r06f/infer_fusion_check.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r06f/pytest.txt 2>&1
grep -E "passed|failed" gpurun_out/r06f/pytest.txt
