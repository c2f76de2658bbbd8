#!/bin/bash
# round 4, step n: every full-batch variable's error over the oracle's own fp32-vs-fp64 deviation, both fp32 modes (restore the
# general 2x rule of tests/test_tight_gpu.py and list the outliers explicitly)
cd /root/repo; mkdir -p gpurun_out/r04n; rm -f gpurun_out/r04n/ratios.jsonl
EG_TIGHT_RATIOS=gpurun_out/r04n/ratios.jsonl timeout 1500 python -m pytest tests/test_tight_gpu.py::test_full_batch_fp32_substep_against_fp64_oracle tests/test_x3_gpu.py::test_x3_full_batch_substep_against_fp64_oracle -q -p no:cacheprovider 2>&1 | tail -3
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r04n/ratios.jsonl')]
rows.sort(key=lambda r:-r['ratio'])
for r in rows[:25]: print('%-6s %-15s %-70s err %.2e self %.2e ratio %.2f tol %.2e'%(r['mode'],r['substep'],r['variable'][:70],r['err'],r['self_dev'],r['ratio'],r['tol']))
PY
