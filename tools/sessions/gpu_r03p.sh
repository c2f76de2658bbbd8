#!/bin/bash
OUT=gpurun_out/r03p
mkdir -p $OUT
L="edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip_bis_190ae2f.so edgegan_amd/libedgegan_hip.so"
EG_LANES=7 python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 --dtype bf16 > $OUT/lanes7_256_bf16.txt 2>&1
EG_LANES=0 python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 --dtype bf16 > $OUT/lanes0_256_bf16.txt 2>&1
echo lanes7; cat $OUT/lanes7_256_bf16.txt; echo lanes0; cat $OUT/lanes0_256_bf16.txt
