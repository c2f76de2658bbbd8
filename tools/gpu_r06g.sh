#!/bin/bash
# inference: calls in flight x hardware queues
cd /root/repo
mkdir -p gpurun_out/r06g
for Q in 4 8; do for S in 4 8; do
  GPU_MAX_HW_QUEUES=$Q EG_INFER_SLOTS=$S timeout 200 python bench.py --mode infer --steps 400 --warmup 20 --profile-steps 0 --no-live-traffic > gpurun_out/r06g/infer_q${Q}_s$S.json 2> gpurun_out/r06g/infer_q${Q}_s$S.err
  python -c "import json;d=json.load(open('gpurun_out/r06g/infer_q${Q}_s$S.json'));print('hw queues $Q slots $S', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done; done
timeout 300 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "in_flight or inference" 2>&1 | tail -3
