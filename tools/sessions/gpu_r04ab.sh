#!/bin/bash
# x3h: fragment reads spread over the six products (one group per product) vs the burst schedule of rounds 2-3 (lib_burst.so)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04ab; mkdir -p $O
timeout 900 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle -m gpu -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for lib in libedgegan_hip.so lib_burst.so; do
  echo "== $lib" | tee -a $O/microbench.txt
  EG_LIB_PATH=edgegan_amd/$lib DTYPE=f32x3 SHAPES="cls Conv_2,E.res,D.conv" python tools/gpu_microbench.py 2>/dev/null | grep -v "^shape" | tee -a $O/microbench.txt | cut -c1-100
done
python tools/gpu_ab.py edgegan_amd/libedgegan_hip.so edgegan_amd/lib_burst.so 3 | tee $O/ab_spread_reads.txt
