#!/bin/bash
# round 3, second contact: the remaining full-batch oracle tests (f32x3), lanes/leaf bitwise tests, leaf-stream A/B
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_step_gpu.py -x -q -k "lanes or train_step_equals or rccl" --durations=10 > $OUT/pytest_lanes.txt 2>&1; echo "lanes rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_dp_gpu.py -x -q > $OUT/pytest_dp.txt 2>&1; echo "dp rc=$?" >> $OUT/rc.txt
python tools/gpu_ab_env.py EG_LANES=7 EG_LANES=15 4 > $OUT/ab_leaf.txt 2>&1
python tools/gpu_ab_env.py EG_LANES=7 EG_LANES=15 3 --dtype bf16 > $OUT/ab_leaf_bf16.txt 2>&1
EG_TIGHT_REPORT=1 python -m pytest tests/test_tight_gpu.py tests/test_x3_gpu.py -q -s -k "full_batch" --durations=20 > $OUT/pytest_fullbatch.txt 2>&1; echo "fullbatch rc=$?" >> $OUT/rc.txt
python bench.py --mode loop --steps 30 > $OUT/bench_loop.json 2> $OUT/bench_loop.err; echo "loop rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt $OUT/ab_leaf.txt $OUT/ab_leaf_bf16.txt
for f in $OUT/pytest_*.txt; do tail -n 4 $f; done
cat $OUT/bench_loop.json
