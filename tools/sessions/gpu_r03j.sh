#!/bin/bash
OUT=gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_step_gpu.py tests/test_x3_gpu.py -x -q -k "train_step_equals or substeps_14 or odd_batch or network_forwards or margin_selected" > $OUT/pytest_step.txt 2>&1; echo "step rc=$?" >> $OUT/rc.txt
python tools/gpu_ab.py edgegan_amd/libedgegan_hip_prev.so edgegan_amd/libedgegan_hip.so 4 > $OUT/ab_elem_fusion.txt 2>&1
EG_TIGHT_REPORT=1 python -m pytest tests/test_tight_gpu.py tests/test_x3_gpu.py -q -s -k "full_batch and d_optim2" > $OUT/pytest_fullbatch.txt 2>&1; echo "fullbatch rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt $OUT/ab_elem_fusion.txt
for f in $OUT/pytest_*.txt; do tail -n 3 $f; done
