#!/bin/bash
# round 4, step h: the 4x4 stride-2 forward convolutions in space-to-depth form on the plane-fed kernel
cd /root/repo; mkdir -p gpurun_out/r04h; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bf16_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04h/pytest_bf16.txt
tail -4 gpurun_out/r04h/pytest_bf16.txt | cut -c1-250
DTYPE=bf16 SHAPES="cls Conv_2,E.res,D.conv" python tools/gpu_microbench.py 2>&1 | grep -E "conv_bf16p|conv_gemm" > gpurun_out/r04h/micro.txt
EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so EG_BF16P_NO_BIG=1 DTYPE=bf16 SHAPES="cls Conv_2" python tools/gpu_microbench.py 2>&1 | grep -E "conv_bf16p|conv_gemm" > gpurun_out/r04h/micro_nobig.txt
cat gpurun_out/r04h/micro.txt; echo; cat gpurun_out/r04h/micro_nobig.txt
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04h/bench_bf16.json 2> gpurun_out/r04h/bench.err
cut -c1-200 gpurun_out/r04h/bench_bf16.json
python tools/layer_table.py gpurun_out/r04h/layers_bf16.csv --dtype bf16 2> gpurun_out/r04h/layers.err
