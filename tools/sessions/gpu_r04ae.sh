#!/bin/bash
# bf16p: how much of the filter-DMA cost is waiting for a tile requested one step earlier?  (timing build: no vmcnt wait at the step top)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04ae; mkdir -p $O
for lib in libedgegan_hip.so lib_dbgp64.so; do
  echo "== $lib" | tee -a $O/microbench.txt
  EG_LIB_PATH=edgegan_amd/$lib DTYPE=bf16 SHAPES="cls Conv_2,E.res 3x3 REFLECT 128,D.conv3,D.conv4 patch" python tools/gpu_microbench.py 2>/dev/null | grep "conv_bf16p" | tee -a $O/microbench.txt | cut -c1-110
done
