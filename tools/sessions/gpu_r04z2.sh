#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04z2; mkdir -p $O
timeout 1500 python -m pytest tests/test_tight_gpu.py "tests/test_x3_gpu.py::test_x3_full_batch_substep_against_fp64_oracle" -m gpu -q > $O/pytest_tight.txt 2>&1
tail -5 $O/pytest_tight.txt
