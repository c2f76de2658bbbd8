#!/bin/bash
# Timing-experiment builds of the halo kernel (EG_X3H_DBG bit mask, wrong results by design): edgegan_amd/lib_dbg<N>.so
cd $(dirname $0)/../edgegan_amd/csrc
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -DEG_X3H_DBG=$n -c k_gemm_x3h.hip -o /tmp/x3h_dbg$n.o || exit 1
  objs=$(ls build/*.o | grep -v k_gemm_x3h.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_dbg$n.so $objs /tmp/x3h_dbg$n.o -ldl || exit 1
done
