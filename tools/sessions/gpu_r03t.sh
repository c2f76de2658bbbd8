#!/bin/bash
OUT=gpurun_out/r03t
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $OUT/rc.txt
for m in 15 7 11 13 14 3; do python tools/gpu_ab_env.py EG_LANES=$m EG_LANES=15 2 2>&1 | head -1 >> $OUT/mask_sweep.txt; done
cat $OUT/rc.txt $OUT/mask_sweep.txt; tail -25 $OUT/pytest_gpu.txt
