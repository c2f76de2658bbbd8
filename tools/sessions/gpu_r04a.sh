#!/bin/bash
# round 4, first session step: the S2D one-plane fix (dead look-ahead read kept live until its wait) -- tools/gpu_s2d_check.py in
# both modes, the op tests that cover it, then this box's baseline bench lines (f32x3, bf16) for the A/B work that follows
cd /root/repo; mkdir -p gpurun_out/r04a
python tools/gpu_s2d_check.py > gpurun_out/r04a/s2d_check.txt 2>&1
cat gpurun_out/r04a/s2d_check.txt | cut -c1-150
timeout 900 python -m pytest tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances -x -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r04a/pytest_ops.txt
cat gpurun_out/r04a/pytest_ops.txt
python bench.py --no-cpu-baseline > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04a/bench_bf16.json 2>> gpurun_out/r04a/bench.err
cut -c1-300 gpurun_out/r04a/bench.json gpurun_out/r04a/bench_bf16.json
