#!/bin/bash
# round 6, second GPU session: whole GPU suite on the 6F-F0 gradient-penalty branch + inference slots; inference throughput by slot count
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
for S in 1 2 4; do
  EG_INFER_SLOTS=$S timeout 200 python bench.py --mode infer --steps 300 --warmup 20 --profile-steps 0 --no-live-traffic > gpurun_out/r06b/infer_slots$S.json 2> gpurun_out/r06b/infer_slots$S.err
  python -c "import json;d=json.load(open('gpurun_out/r06b/infer_slots$S.json'));print('slots $S', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
( time timeout 400 python bench.py --no-cpu-baseline ) > gpurun_out/r06b/bench.json 2> gpurun_out/r06b/bench.err
python -c "import json;d=json.load(open('gpurun_out/r06b/bench.json'));print(d['value'], d['roofline']['all_gemm_kernels'], d['bf16_mode']['value'], d['config1_infer_b64']['value'])"
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06b/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r06b/pytest_gpu.txt
