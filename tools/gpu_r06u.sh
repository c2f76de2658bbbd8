#!/bin/bash
# round 6, second session: register-resident InstanceNorm backward with all loads issued up front (HEAD) against the per-element form (prev)
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -3 $OUT/pytest_ops.txt
timeout 900 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "substeps_14 or single_class or train_step_equals" > $OUT/pytest_step.txt 2>&1
tail -3 $OUT/pytest_step.txt
for L in prev cur; do
  LIB=edgegan_amd/libedgegan_hip_prev.so; [ $L = cur ] && LIB=edgegan_amd/libedgegan_hip.so
  bash tools/gpu_trace.sh $OUT/kernel_stats_bf16_$L.csv $LIB --dtype bf16 --no-side-legs --no-live-traffic
  bash tools/gpu_trace.sh $OUT/kernel_stats_b4_$L.csv $LIB --batch 4 --classes 0 --no-side-legs --no-live-traffic
done
for L in prev cur; do echo "== $L"; grep -h "in_bwd_vr\|minmax\|TOTAL" $OUT/kernel_stats_bf16_$L.csv | cut -c1-40,150-220; grep -h "in_bwd_vr\|minmax\|TOTAL" $OUT/kernel_stats_b4_$L.csv | cut -c1-40,150-220; done
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1; cat $OUT/ab_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype f32x3 > $OUT/ab_f32x3.txt 2>&1; cat $OUT/ab_f32x3.txt
