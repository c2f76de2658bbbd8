#!/bin/bash
# round 3, first contact: the new / changed gpu tests, one bench line, the per-layer table (baseline for the kernel work)
OUT=gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_dp_gpu.py -x -q --durations=20 > $OUT/pytest_dp.txt 2>&1; echo "dp rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_step_gpu.py -x -q -k "branch_lanes or concurrent_lanes or rccl" --durations=20 > $OUT/pytest_lanes.txt 2>&1; echo "lanes rc=$?" >> $OUT/rc.txt
EG_TIGHT_REPORT=1 python -m pytest tests/test_tight_gpu.py tests/test_x3_gpu.py -x -q -s -k "full_batch" --durations=20 > $OUT/pytest_fullbatch.txt 2>&1; echo "fullbatch rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_bf16_gpu.py -x -q -k "full_size" --durations=20 > $OUT/pytest_bf16_full.txt 2>&1; echo "bf16 rc=$?" >> $OUT/rc.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python tools/layer_table.py $OUT/layers.csv 2> $OUT/layers.err
cat $OUT/rc.txt
tail -5 $OUT/pytest_*.txt
