#!/bin/bash
OUT=gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_step_gpu.py -x -q -k "lanes or train_step_equals or substeps_14 or odd_batch or network_forwards" --durations=10 > $OUT/pytest_step.txt 2>&1; echo "step rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?" >> $OUT/rc.txt
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so python tools/gpu_ab_env.py EG_PROJ_FULL_RES=1 EG_LANES=15 4 > $OUT/ab_proj.txt 2>&1
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so python tools/gpu_ab_env.py EG_PROJ_FULL_RES=1 EG_LANES=15 3 --dtype bf16 > $OUT/ab_proj_bf16.txt 2>&1
EG_TIGHT_REPORT=1 python -m pytest tests/test_tight_gpu.py tests/test_x3_gpu.py -q -s -k "full_batch and (g_optim_u or d_optim2)" > $OUT/pytest_fullbatch.txt 2>&1; echo "fullbatch rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt $OUT/ab_proj.txt $OUT/ab_proj_bf16.txt
for f in $OUT/pytest_*.txt; do tail -n 4 $f; done
