#!/bin/bash
# round 4, step e: plane-fed bf16 kernel with the statically unrolled K loop: parity (op tests), microbench + limiter variants
cd /root/repo; mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle -x -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r04e/pytest_ops.txt
tail -5 gpurun_out/r04e/pytest_ops.txt | cut -c1-250
for v in base 1 4 8 6 22 30 63; do
  if [ $v = base ]; then lib=edgegan_amd/libedgegan_hip.so; else lib=edgegan_amd/lib_dbgp$v.so; fi
  echo "== variant $v" >> gpurun_out/r04e/limiter.txt
  EG_LIB_PATH=$lib DTYPE=bf16 SHAPES="cls Conv_2,E.res,D.conv" python tools/gpu_microbench.py 2>&1 | grep -E "conv_bf16p|conv_gemm" >> gpurun_out/r04e/limiter.txt
done
cat gpurun_out/r04e/limiter.txt
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04e/bench_bf16.json 2> gpurun_out/r04e/bench.err
cut -c1-200 gpurun_out/r04e/bench_bf16.json
