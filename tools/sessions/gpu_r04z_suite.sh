#!/bin/bash
# the whole GPU suite + smoke at the current HEAD
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
