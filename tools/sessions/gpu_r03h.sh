#!/bin/bash
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_x3_gpu.py -x -q -k "ops_at_fp32 or network_forwards or substeps_14" --durations=5 > $OUT/pytest_x3.txt 2>&1; echo "x3 rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_bf16_gpu.py -x -q -k "ops_match or network_forwards" > $OUT/pytest_bf16.txt 2>&1; echo "bf16 rc=$?" >> $OUT/rc.txt
python -m pytest tests/test_ops_gpu.py -x -q > $OUT/pytest_ops.txt 2>&1; echo "ops rc=$?" >> $OUT/rc.txt
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so python tools/gpu_ab_env.py EG_NO_S2D=1 EG_LANES=15 4 > $OUT/ab_s2d.txt 2>&1
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so python tools/gpu_ab_env.py EG_NO_S2D=1 EG_LANES=15 3 --dtype bf16 > $OUT/ab_s2d_bf16.txt 2>&1
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so DTYPE=f32x3 SHAPES="D.conv" python tools/gpu_microbench.py > $OUT/micro_s2d.txt 2>&1
EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so EG_NO_S2D=1 DTYPE=f32x3 SHAPES="D.conv" python tools/gpu_microbench.py > $OUT/micro_nos2d.txt 2>&1
cat $OUT/rc.txt $OUT/ab_s2d.txt $OUT/ab_s2d_bf16.txt
for f in $OUT/pytest_*.txt; do tail -n 4 $f; done
grep "fwd\|x3h\|x3pp" $OUT/micro_s2d.txt | head -12; echo ---; grep "x3h\|x3pp" $OUT/micro_nos2d.txt | head -12
