#!/bin/bash
OUT=gpurun_out/r03v
mkdir -p $OUT
export EG_LIB_PATH=$PWD/edgegan_amd/libedgegan_hip_exp.so
python tools/gpu_ab_env.py EG_SS6_LANE=1 EG_SS6_LANE=2 3 > $OUT/ab_ss6_lane2.txt 2>&1
python tools/gpu_ab_env.py EG_SS6_LANE=1 EG_SS6_LANE=3 3 > $OUT/ab_ss6_lane3.txt 2>&1
python tools/gpu_ab_env.py EG_SS6_LANE=1 EG_SS6_LANE=2 2 --dtype bf16 > $OUT/ab_ss6_lane2_bf16.txt 2>&1
python tools/gpu_ab_env.py EG_SS6_LANE=1 EG_SS6_LANE=2 2 --height 256 --width 512 --batch 16 --steps 4 > $OUT/ab_ss6_lane2_256.txt 2>&1
cat $OUT/*.txt
