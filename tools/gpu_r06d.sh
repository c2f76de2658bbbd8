#!/bin/bash
# round 6, fourth GPU session: generator / encoder head fusions, plane-only InstanceNorm outputs in forward-only bf16 graphs;
# the whole GPU suite (flip tolerances removed: both fp32 modes evaluate the oracle live; bf16 digests at 256x512)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06d
timeout 600 python tools/gpu_infer_fusion_check.py > gpurun_out/r06d/infer_fusion_check.txt 2>&1
tail -9 gpurun_out/r06d/infer_fusion_check.txt
for S in 1 4; do
  EG_INFER_SLOTS=$S timeout 200 python bench.py --mode infer --steps 300 --warmup 20 --profile-steps 0 --no-live-traffic > gpurun_out/r06d/infer_slots$S.json 2> gpurun_out/r06d/infer_slots$S.err
  python -c "import json;d=json.load(open('gpurun_out/r06d/infer_slots$S.json'));print('slots $S', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
rm -rf /tmp/inf0
EG_LANES=0 timeout 300 rocprofv3 --kernel-trace -d /tmp/inf0 -o t -- python bench.py --mode infer --steps 20 --warmup 5 --profile-steps 0 --no-live-traffic > gpurun_out/r06d/infer_lanes0.json 2> gpurun_out/r06d/infer_lanes0.err
python tools/infer_timeline.py $(find /tmp/inf0 -name "*.db" | head -1) 0 5 > gpurun_out/r06d/infer_timeline_lanes0.txt 2>&1
head -4 gpurun_out/r06d/infer_timeline_lanes0.txt | tail -2
( time timeout 400 python bench.py ) > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err
python -c "import json;d=json.load(open('gpurun_out/r06d/bench.json'));print(d['value'], d['bf16_mode']['value'], d['config4_256x512_b16_bf16']['value'], d['config1_infer_b64']['value'], d['bench_seconds'])"
EG_FLIP_REPORT=1 timeout 1800 python -m pytest tests -x -q -m gpu -s > gpurun_out/r06d/pytest_gpu.txt 2>&1
grep -E "passed|failed|flip proof" gpurun_out/r06d/pytest_gpu.txt | tail -12
