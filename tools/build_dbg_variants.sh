#!/bin/bash
# Experiment builds of the halo kernel: edgegan_amd/lib_<tag>.so for each "tag=flags" argument, e.g.
#   tools/build_dbg_variants.sh dbg32=-DEG_X3H_DBG=32 agpr=-DEG_X3H_AGPR=1
cd $(dirname $0)/../edgegan_amd/csrc
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $flags -c k_gemm_x3h.hip -o /tmp/x3h_$tag.o || exit 1
  objs=$(ls build/*.o | grep -v k_gemm_x3h.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$tag.so $objs /tmp/x3h_$tag.o -ldl || exit 1
done
