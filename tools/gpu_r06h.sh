#!/bin/bash
# the training iteration as a captured HIP graph: equality test, then iteration time with / without at small and full batch
cd /root/repo
mkdir -p gpurun_out/r06h
timeout 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "graph_replayed or equals_sequential or concurrent_lanes" > gpurun_out/r06h/pytest.txt 2>&1
tail -5 gpurun_out/r06h/pytest.txt
for cfgs in "--batch 4 --classes 0" "--batch 1" "--batch 16" "--batch 64" "--batch 64 --dtype bf16"; do for G in 0 1; do
  out=$(EG_GRAPH=$G timeout 300 python bench.py $cfgs --steps 40 --warmup 6 --no-cpu-baseline --no-alt --no-side-legs --no-live-traffic --profile-steps 0 2>&1 | tail -1)
  echo "EG_GRAPH=$G $cfgs -> $(echo "$out" | python -c "import sys,json; l=sys.stdin.read().strip().splitlines()[-1]
try:
    d=json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms/iteration, host enqueue', d['host_enqueue_ms_per_step'], d.get('losses_finite'))
except Exception as e: print('FAIL', l[:300])")" | tee -a gpurun_out/r06h/graph_ab.txt
done; done
