#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03x
export EG_LIB_PATH=/root/repo/edgegan_amd/libedgegan_hip_exp.so
for d in 0 32 1 33 0; do
  echo "== EG_WG_DBG=$d" >> gpurun_out/r03x/wgrad_x3r_limiter.txt
  DTYPE=f32x3 SHAPES='cls Conv_2 u1,cls Conv_2 u3,D.conv3,D.conv4 patch' EG_WG_DBG=$d timeout 300 python tools/gpu_microbench.py 2>&1 | grep -i "wgrad" >> gpurun_out/r03x/wgrad_x3r_limiter.txt
done
cat gpurun_out/r03x/wgrad_x3r_limiter.txt
