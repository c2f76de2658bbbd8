#!/bin/bash
OUT=gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
export EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so
python -m pytest tests/test_x3_gpu.py -x -q -k "ops_at_fp32" > $OUT/pytest_x3ops.txt 2>&1; echo "x3ops rc=$?" >> $OUT/rc.txt
DTYPE=f32x3 python tools/gpu_microbench.py > $OUT/micro_tr.txt 2>&1
EG_NO_WGRAD_TR=1 DTYPE=f32x3 python tools/gpu_microbench.py > $OUT/micro_notr.txt 2>&1
python tools/gpu_ab_env.py EG_NO_WGRAD_TR=1 EG_LANES=15 4 > $OUT/ab_wgrad_tr.txt 2>&1
cat $OUT/rc.txt $OUT/ab_wgrad_tr.txt; tail -5 $OUT/pytest_x3ops.txt
grep wgrad $OUT/micro_tr.txt; echo ---; grep wgrad $OUT/micro_notr.txt
