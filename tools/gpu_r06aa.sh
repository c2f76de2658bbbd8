#!/bin/bash
# round 6, second session: chunked two-stage InstanceNorm on large maps (HEAD) against the one-block-per-(image, channel group) form
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06aa}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_bf16_gpu.py tests/test_x3_gpu.py -q -m gpu -x -k "high_resolution or 256 or full_size or config4" > $OUT/pytest_256.txt 2>&1
tail -3 $OUT/pytest_256.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -2 $OUT/pytest_ops.txt
for L in prev cur; do
  LIB=edgegan_amd/libedgegan_hip_prev.so; [ $L = cur ] && LIB=edgegan_amd/libedgegan_hip.so
  bash tools/gpu_trace.sh $OUT/kernel_stats_256_bf16_$L.csv $LIB --dtype bf16 --height 256 --width 512 --batch 16 --steps 4 --warmup 2 --no-side-legs --no-live-traffic
done
for L in prev cur; do echo "== $L"; grep -h "finalize\|TOTAL" $OUT/kernel_stats_256_bf16_$L.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:44].ljust(44), r[1].rjust(5), r[3].rjust(10), r[2].rjust(12))"; done
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype bf16 --height 256 --width 512 --batch 16 --steps 5 | tee $OUT/ab_256_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so --rounds 2 --dtype f32x3 --height 256 --width 512 --batch 16 --steps 4 | tee $OUT/ab_256_f32x3.txt
