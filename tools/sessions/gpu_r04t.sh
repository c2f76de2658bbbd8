#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
timeout 900 python -m pytest tests/test_x3_gpu.py::test_x3_ops_at_fp32_tolerances -m gpu -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
python tools/layer_table.py $O/layers.csv > $O/layers.txt 2>&1
grep "k7\|k5s2" $O/layers.csv | cut -c1-200
