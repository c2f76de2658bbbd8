#!/bin/bash
# round 6, first GPU session: new-flag tests, the default bench with its side legs, inference timelines
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
timeout 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "non_default or single_class or identity" > gpurun_out/r06a/pytest_flags.txt 2>&1
tail -3 gpurun_out/r06a/pytest_flags.txt
( time timeout 400 python bench.py ) > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
tail -5 gpurun_out/r06a/bench.err
for L in 15 0; do
  rm -rf /tmp/inf$L
  EG_LANES=$L timeout 300 rocprofv3 --kernel-trace -d /tmp/inf$L -o t -- python bench.py --mode infer --steps 20 --warmup 5 --profile-steps 0 --no-live-traffic > gpurun_out/r06a/infer_lanes$L.json 2> gpurun_out/r06a/infer_lanes$L.err
  DB=$(find /tmp/inf$L -name "*.db" | head -1)
  python tools/infer_timeline.py $DB 0 5 > gpurun_out/r06a/infer_timeline_lanes$L.txt 2>&1
done
head -3 gpurun_out/r06a/infer_timeline_lanes15.txt
