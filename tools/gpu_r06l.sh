#!/bin/bash
# round 6, second session: rows_linear + 16-wave min-max gate (lib B) against the element-wise pass 1 (lib A); concurrent and
# serialised kernel traces of both arithmetic modes with the solo-time table (tools/trace_overlap.py)
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -3 $OUT/pytest_ops.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_bf16_gpu.py -q -m gpu -x -k "substeps or train_step_equals or inference" > $OUT/pytest_step.txt 2>&1
tail -3 $OUT/pytest_step.txt
for D in bf16 f32x3; do
  for LANES in 15 0; do
    rm -rf /tmp/tr_${D}_$LANES
    EG_LANES=$LANES timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_${D}_$LANES -o trace -- \
      python bench.py --dtype $D --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt --no-side-legs --no-live-traffic > $OUT/tr_${D}_$LANES.bench.json 2> $OUT/tr_${D}_$LANES.err
  done
  C=$(find /tmp/tr_${D}_15 -name '*.db' | head -1); S=$(find /tmp/tr_${D}_0 -name '*.db' | head -1)
  MS=$(python -c "import json;print(json.loads(open('$OUT/tr_${D}_15.bench.json').read().strip().splitlines()[-1])['ms_per_step']*10)")
  python tools/trace_overlap.py $C $S 45 $MS > $OUT/overlap_$D.txt 2>&1
  python tools/rocprof_summary.py $S $OUT/kernel_stats_$D.csv 13
done
grep -h "minmax\|rows_linear\|conv_gemm_kernel<64, 64\|TOTAL" $OUT/kernel_stats_bf16.csv | cut -c1-60,100-200
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_A.so edgegan_amd/libedgegan_hip_B.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1
cat $OUT/ab_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_A.so edgegan_amd/libedgegan_hip_B.so --rounds 3 --dtype f32x3 > $OUT/ab_f32x3.txt 2>&1
cat $OUT/ab_f32x3.txt
for R in 1 2 3; do for L in A B; do
  echo -n "infer $L "
  EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_$L.so python bench.py --mode infer --profile-steps 0 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $OUT/ab_infer.txt
