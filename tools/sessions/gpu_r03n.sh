#!/bin/bash
OUT=gpurun_out/r03n
mkdir -p $OUT
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip.so --rounds 4 > $OUT/ab_r02_r03.txt 2>&1
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip.so --rounds 3 --dtype bf16 > $OUT/ab_r02_r03_bf16.txt 2>&1
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip.so --rounds 3 --height 256 --width 512 --batch 16 --steps 4 > $OUT/ab_r02_r03_256.txt 2>&1
python tools/gpu_ab_rounds.py edgegan_amd/libedgegan_hip_r02.so edgegan_amd/libedgegan_hip.so --rounds 2 --height 256 --width 512 --batch 16 --steps 4 --dtype bf16 > $OUT/ab_r02_r03_256_bf16.txt 2>&1
cat $OUT/*.txt
