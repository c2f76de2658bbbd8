#!/bin/bash
# generalised packed-K thin kernel (dgrad, 7x7, 5x5s2, 1x1, any N): parity in the three modes, benches, layer table
set -x
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04r; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle tests/test_bf16_gpu.py::test_bf16_network_forwards -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>> $O/bench.err; tail -c 300 $O/bench_bf16.json
timeout 600 python tools/layer_table.py $O/layers.csv > $O/layers.txt 2>&1; grep -i "conv_pk\|smallc\|scatter" $O/layers.csv | head -60
