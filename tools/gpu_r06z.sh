#!/bin/bash
# round 6, final profile set at HEAD: both arithmetic modes (PMC passes, bench lines, serialised traces, per-layer tables),
# the inference timeline and fusion check, smoke, the whole GPU suite
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/gpu_round_profile.sh $TAG > $OUT/round_profile.log 2>&1
bash tools/gpu_round_profile_bf16.sh $TAG > $OUT/round_profile_bf16.log 2>&1
rm -rf /tmp/inf0
EG_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/inf0 -o t -- python bench.py --mode infer --steps 20 --warmup 5 --profile-steps 0 --no-live-traffic > $OUT/kernel_stats_infer.bench.json 2> $OUT/prof_infer.err
DB=$(find /tmp/inf0 -name "*.db" | head -1)
python tools/infer_timeline.py $DB 0 5 > $OUT/infer_timeline.txt 2>&1
python tools/rocprof_summary.py $DB $OUT/kernel_stats_infer.csv 25
timeout 600 python tools/gpu_infer_fusion_check.py > $OUT/infer_fusion_check.txt 2>&1
EG_FLIP_REPORT=1 timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/pytest_gpu_full.txt 2>&1
grep -E "passed|failed|flip proof" $OUT/pytest_gpu_full.txt | cut -c1-300 > $OUT/pytest_gpu.txt
tail -3 $OUT/pytest_gpu.txt
python -c "
import json
d=json.load(open('$OUT/bench.json'))
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
for k in ('bf16_mode','config4_256x512_b16_bf16','config1_infer_b64'):
    l=d[k]; print(k, l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['frac'], l['roofline'].get('traffic'))
print('cpu', d['cpu_baseline']['value'], 'seconds', d['bench_seconds'])
"
cat $OUT/smoke.txt | tail -2
ls $OUT | head -50
