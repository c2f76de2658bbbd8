#!/bin/bash
# round 6: bf16 mode, 1x1 convolutions on the 64 x 64-tile gather kernel instead of a < 256-block ping-pong launch (lib A: -DEG_PW_SMALL=1)
cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/r06ai
mkdir -p $OUT
EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_A.so timeout 900 python -m pytest tests/test_bf16_gpu.py -q -m gpu -x > $OUT/pytest_bf16_A.txt 2>&1
tail -3 $OUT/pytest_bf16_A.txt
for L in base A; do
  bash tools/gpu_trace.sh $OUT/kernel_stats_bf16_$L.csv edgegan_amd/libedgegan_hip_$L.so --dtype bf16 > $OUT/trace_$L.log 2>&1
  grep -h "x3pp_kernel<\|conv_gemm_bf16_kernel\|TOTAL" $OUT/kernel_stats_bf16_$L.csv | cut -c1-90,100-200
  echo ---
done
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_base.so edgegan_amd/libedgegan_hip_A.so --rounds 3 --dtype bf16 > $OUT/ab_bf16.txt 2>&1
cat $OUT/ab_bf16.txt
