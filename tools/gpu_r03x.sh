#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
for d in 15 143 271 399 398 384; do
  echo "== EG_SMALLC_DBG=$d" >> gpurun_out/r03x/smallc_limiter3.txt
  DTYPE=f32x3 SHAPES='cls Conv_1 u1,D.conv0,cls h0' EG_SMALLC_DBG=$d timeout 300 python tools/gpu_microbench.py 2>&1 | grep -i "smallc" >> gpurun_out/r03x/smallc_limiter3.txt
done
cat gpurun_out/r03x/smallc_limiter3.txt
