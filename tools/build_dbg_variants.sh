#!/bin/bash
# Experiment builds of one kernel file: edgegan_amd/lib_<tag>.so for each "tag=flags" argument (select with EG_LIB_PATH), e.g.
#   tools/build_dbg_variants.sh k_gemm_x3h dbg32=-DEG_X3H_DBG=32 agpr=-DEG_X3H_AGPR=1
#   tools/build_dbg_variants.sh k_conv_bf16p nobar=-DEG_BF16P_DBG=1
cd $(dirname $0)/../edgegan_amd/csrc
src=$1; shift
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $flags -c $src.hip -o /tmp/${src}_$tag.o || exit 1
  objs=$(ls build/*.o | grep -v "/$src.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$tag.so $objs /tmp/${src}_$tag.o -ldl || exit 1
done
