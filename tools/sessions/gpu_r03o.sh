#!/bin/bash
OUT=gpurun_out/r03o
mkdir -p $OUT
L="edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip_bis_190ae2f.so edgegan_amd/libedgegan_hip_bis_64ce266.so edgegan_amd/libedgegan_hip_bis_8af9a1c.so edgegan_amd/libedgegan_hip_bis_2ce770c.so edgegan_amd/libedgegan_hip.so"
python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 --dtype bf16 > $OUT/bisect_256_bf16.txt 2>&1
python tools/gpu_ab_rounds.py $L --rounds 2 --height 256 --width 512 --batch 16 --steps 4 > $OUT/bisect_256.txt 2>&1
cat $OUT/*.txt
