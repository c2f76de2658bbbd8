#!/bin/bash
# round 6, third GPU session: forward-only fusions -- equality check, inference timeline and throughput, the inference / step tests
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 600 python tools/gpu_infer_fusion_check.py > gpurun_out/r06c/infer_fusion_check.txt 2>&1
tail -8 gpurun_out/r06c/infer_fusion_check.txt
for S in 1 2 4; do
  EG_INFER_SLOTS=$S timeout 200 python bench.py --mode infer --steps 300 --warmup 20 --profile-steps 0 --no-live-traffic > gpurun_out/r06c/infer_slots$S.json 2> gpurun_out/r06c/infer_slots$S.err
  python -c "import json;d=json.load(open('gpurun_out/r06c/infer_slots$S.json'));print('slots $S', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
rm -rf /tmp/inf0
EG_LANES=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/inf0 -o t -- python bench.py --mode infer --steps 20 --warmup 5 --profile-steps 0 --no-live-traffic > gpurun_out/r06c/infer_lanes0.json 2> gpurun_out/r06c/infer_lanes0.err
python tools/infer_timeline.py $(find /tmp/inf0 -name "*.db" | head -1) 0 5 > gpurun_out/r06c/infer_timeline_lanes0.txt 2>&1
head -4 gpurun_out/r06c/infer_timeline_lanes0.txt | tail -2
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_golden_gpu.py tests/test_bf16_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/r06c/pytest.txt 2>&1
tail -5 gpurun_out/r06c/pytest.txt
