#!/bin/bash
OUT=gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
/tmp/tr_probe > $OUT/tr_probe.txt 2>&1 || (hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe > $OUT/tr_probe.txt 2>&1)
python -m pytest tests/test_step_gpu.py -x -q -k "lanes or train_step_equals or rccl or substeps_14 or odd_batch" --durations=10 > $OUT/pytest_step.txt 2>&1; echo "step rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_dp_gpu.py -x -q > $OUT/pytest_dp.txt 2>&1; echo "dp rc=$?" >> $OUT/rc.txt
python tools/gpu_ab_env.py EG_LANES=15 EG_LANES=31 4 > $OUT/ab_g2.txt 2>&1
python tools/gpu_ab_env.py EG_CLS_HALVES=0 EG_CLS_HALVES=1 4 > $OUT/ab_halves.txt 2>&1
python tools/gpu_ab_env.py EG_CLS_HALVES=0 EG_CLS_HALVES=1 3 --dtype bf16 > $OUT/ab_halves_bf16.txt 2>&1
EG_TIGHT_REPORT=1 python -m pytest tests/test_tight_gpu.py tests/test_x3_gpu.py -q -s -k "full_batch and (g_optim_u or d_optim2)" --durations=20 > $OUT/pytest_fullbatch.txt 2>&1; echo "fullbatch rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt $OUT/ab_g2.txt $OUT/ab_halves.txt $OUT/ab_halves_bf16.txt
for f in $OUT/pytest_*.txt; do tail -n 4 $f; done
head -20 $OUT/tr_probe.txt
