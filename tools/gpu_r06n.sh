#!/bin/bash
# round 6, second session: E_norm / G_norm / D_norm = 'batch' -- op parity (bn_act, bn_tangent) and the sub-step tests
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-r06n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -x > $OUT/pytest_ops.txt 2>&1
tail -25 $OUT/pytest_ops.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_step_gpu.py -q -m gpu -k "batch_norm_flags" > $OUT/pytest_bn.txt 2>&1
tail -60 $OUT/pytest_bn.txt | cut -c1-300
