#!/bin/bash
# packed thin filter-gradient kernel: op parity in the three modes, then layer tables
set -x
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04q; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances tests/test_bf16_gpu.py::test_bf16_ops_match_rounded_operand_oracle -m gpu -x -q > $O/pytest_ops.txt 2>&1
tail -15 $O/pytest_ops.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>> $O/bench.err; tail -c 300 $O/bench_bf16.json
timeout 600 python tools/layer_table.py $O/layers.csv > $O/layers.txt 2>&1; grep -i "wgrad_pk\|flatm" $O/layers.csv | head -40
