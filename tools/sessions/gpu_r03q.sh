#!/bin/bash
OUT=gpurun_out/r03q
mkdir -p $OUT
L="edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so"
python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 --dtype bf16 > $OUT/q_256_bf16.txt 2>&1
python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 > $OUT/q_256.txt 2>&1
python tools/gpu_ab_rounds.py $L --rounds 3 > $OUT/q_64.txt 2>&1
python tools/gpu_ab_rounds.py $L --rounds 3 --dtype bf16 > $OUT/q_64_bf16.txt 2>&1
GPU_MAX_HW_QUEUES=4 python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip.so edgegan_amd/libedgegan_hip_prev.so --rounds 2 > $OUT/q4_64.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip.so edgegan_amd/libedgegan_hip_prev.so --rounds 2 > $OUT/q8_64.txt 2>&1
for f in $OUT/*.txt; do echo $f; cat $f; done
