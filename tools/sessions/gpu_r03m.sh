#!/bin/bash
OUT=gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
export EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so
python -m pytest tests/test_step_gpu.py tests/test_x3_gpu.py tests/test_bf16_gpu.py -x -q -k "substeps_14 or single_class or train_step_equals or bf16_substeps" > $OUT/pytest_step.txt 2>&1; echo "step rc=$?" >> $OUT/rc.txt
python tools/gpu_ab_env.py EG_NO_EPI_ACT=1 EG_LANES=15 4 > $OUT/ab_epi_act.txt 2>&1
python tools/gpu_ab_env.py EG_LANES=7 EG_LANES=15 2 --height 256 --width 512 --batch 16 > $OUT/ab_leaf_256.txt 2>&1
python tools/gpu_ab_env.py EG_NO_S2D=1 EG_PROJ_FULL_RES=1 2 --height 256 --width 512 --batch 16 > $OUT/ab_s2d_proj_256.txt 2>&1
cat $OUT/rc.txt $OUT/ab_epi_act.txt $OUT/ab_leaf_256.txt $OUT/ab_s2d_proj_256.txt; tail -4 $OUT/pytest_step.txt
