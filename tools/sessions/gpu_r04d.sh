#!/bin/bash
# round 4, step d: where the plane-fed bf16 kernel's time goes -- timing-only builds (EG_BF16P_DBG masks, wrong results by design)
# on the classifier's Conv_2 layers (tools/gpu_microbench.py, DTYPE=bf16)
cd /root/repo; mkdir -p gpurun_out/r04d
for v in base 1 2 4 8 16 32 64 6 22 30 63; do
  if [ $v = base ]; then lib=edgegan_amd/libedgegan_hip.so; else lib=edgegan_amd/lib_dbgp$v.so; fi
  echo "== variant $v" >> gpurun_out/r04d/limiter.txt
  EG_LIB_PATH=$lib DTYPE=bf16 SHAPES="cls Conv_2" python tools/gpu_microbench.py 2>&1 | grep -E "conv_bf16p|conv_gemm" >> gpurun_out/r04d/limiter.txt
done
cat gpurun_out/r04d/limiter.txt
