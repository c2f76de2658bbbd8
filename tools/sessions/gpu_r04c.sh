#!/bin/bash
# round 4, step c: first contact of the plane-fed bf16 convolution kernel (k_conv_bf16p.hip, planes by launch_cvt_bf16 on demand)
cd /root/repo; mkdir -p gpurun_out/r04c; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_step_gpu.py::test_pipelined_feed_and_loss_readback_equal_the_synchronous_calls -x -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r04c/pytest_ops.txt
cat gpurun_out/r04c/pytest_ops.txt | cut -c1-250
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04c/bench_bf16.json 2> gpurun_out/r04c/bench.err
cut -c1-200 gpurun_out/r04c/bench_bf16.json; tail -3 gpurun_out/r04c/bench.err
python tools/layer_table.py gpurun_out/r04c/layers_bf16.csv --dtype bf16 2> gpurun_out/r04c/layers.err
grep bf16p gpurun_out/r04c/layers_bf16.csv | cut -c1-200 | head -40
