#!/bin/bash
OUT=gpurun_out/r03u
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $OUT/rc.txt
for i in 1 2 3; do python -m pytest tests/test_step_gpu.py tests/test_bf16_gpu.py -q -k "gradient_probes or full_size" >> $OUT/pytest_repeat.txt 2>&1; echo "repeat $i rc=$?" >> $OUT/rc.txt; done
cat $OUT/rc.txt; tail -22 $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_repeat.txt
