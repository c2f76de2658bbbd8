#!/bin/bash
# round 4, final HEAD: the whole GPU suite, smoke, the bench lines of both modes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04zz; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.txt 2>&1
tail -16 $O/pytest_gpu.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-260 $O/bench.json
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err; cut -c1-260 $O/bench_bf16.json
