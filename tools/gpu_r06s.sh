#!/bin/bash
# round 6, second session: XCD-aware block map in the plane-fed filter-gradient kernel (HEAD) against the dispatch order (noxcd)
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_bf16_gpu.py -q -m gpu -x > $OUT/pytest_bf16.txt 2>&1
tail -3 $OUT/pytest_bf16.txt
bash tools/gpu_trace.sh $OUT/kernel_stats_bf16_noxcd.csv edgegan_amd/libedgegan_hip_noxcd.so --dtype bf16 --no-side-legs --no-live-traffic
bash tools/gpu_trace.sh $OUT/kernel_stats_bf16_xcd.csv edgegan_amd/libedgegan_hip.so --dtype bf16 --no-side-legs --no-live-traffic
grep -h "wgrad_bf16p\|TOTAL" $OUT/kernel_stats_bf16_noxcd.csv | cut -c1-70,70-200
echo ---
grep -h "wgrad_bf16p\|TOTAL" $OUT/kernel_stats_bf16_xcd.csv | cut -c1-70,70-200
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_noxcd.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1
cat $OUT/ab_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_noxcd.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype bf16 --height 256 --width 512 --batch 16 --steps 4 > $OUT/ab_bf16_256.txt 2>&1
cat $OUT/ab_bf16_256.txt
