#!/bin/bash
# round 6, second session: vectorised spectral-norm backward / GP norm / register-tap resize (lib A) against HEAD (base), and the
# conv_scatter operand prefetch depth 4 -> 8 (pd8): op parity first, then a serialised kernel trace in bf16 mode, then in-box A/B
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -3 $OUT/pytest_ops.txt
timeout 900 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "substeps or train_step_equals" > $OUT/pytest_step.txt 2>&1
tail -3 $OUT/pytest_step.txt
for L in base A pd8; do
  rm -rf /tmp/prof_$L
  EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_$L.so EG_LANES=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$L -o trace -- \
    python bench.py --dtype bf16 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt --no-side-legs --no-live-traffic > $OUT/ks_$L.bench.json 2> $OUT/prof_$L.err
  DB=$(find /tmp/prof_$L -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB $OUT/kernel_stats_bf16_$L.csv 13; fi
done
grep -h "sn_bwd\|gp_norm\|resize3\|conv_scatter\|TOTAL" $OUT/kernel_stats_bf16_base.csv | cut -c1-60,100-200
echo ---
grep -h "sn_bwd\|gp_norm\|resize3\|conv_scatter\|TOTAL" $OUT/kernel_stats_bf16_A.csv | cut -c1-60,100-200
echo ---
grep -h "conv_scatter\|TOTAL" $OUT/kernel_stats_bf16_pd8.csv | cut -c1-60,100-200
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_base.so edgegan_amd/libedgegan_hip_A.so edgegan_amd/libedgegan_hip_pd8.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1
cat $OUT/ab_bf16.txt
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_base.so edgegan_amd/libedgegan_hip_A.so edgegan_amd/libedgegan_hip_pd8.so --rounds 3 --dtype f32x3 > $OUT/ab_f32x3.txt 2>&1
cat $OUT/ab_f32x3.txt
for L in base A pd8; do
  echo infer $L
  EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_$L.so python bench.py --mode infer --profile-steps 0 --no-live-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $OUT/ab_infer.txt
