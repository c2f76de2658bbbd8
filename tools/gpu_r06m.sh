#!/bin/bash
# round 6, second session: power probe of the two dominant kernels (same launches, random / constant / zero operands), the
# reworked rows_linear + batched limb split (lib C) against lib B
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for D in bf16 f32x3; do for O in random const zero random; do
  echo "== $D operands $O"
  OPERANDS=$O DTYPE=$D SHAPES="cls Conv_2,E.res 3x3 REFLECT 128" python tools/gpu_microbench.py 2>&1 | grep -v "^shape"
done; done > $OUT/power_probe.txt 2>&1
cat $OUT/power_probe.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -3 $OUT/pytest_ops.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_bf16_gpu.py -q -m gpu -x -k "substeps or train_step_equals or inference" > $OUT/pytest_step.txt 2>&1
tail -3 $OUT/pytest_step.txt
bash tools/gpu_trace.sh $OUT/kernel_stats_bf16_C.csv edgegan_amd/libedgegan_hip_C.so --dtype bf16 --no-side-legs --no-live-traffic
bash tools/gpu_trace.sh $OUT/kernel_stats_f32x3_C.csv edgegan_amd/libedgegan_hip_C.so --no-side-legs --no-live-traffic
grep -h "rows_linear\|split_bf16x3\|TOTAL" $OUT/kernel_stats_bf16_C.csv $OUT/kernel_stats_f32x3_C.csv | cut -c1-60,100-200
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_B.so edgegan_amd/libedgegan_hip_C.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1
cat $OUT/ab_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_B.so edgegan_amd/libedgegan_hip_C.so --rounds 3 --dtype f32x3 > $OUT/ab_f32x3.txt 2>&1
cat $OUT/ab_f32x3.txt
