#!/bin/bash
# the exchange's placement knobs on one device (masked communication stream, channel cap) + the DP tests
cd /root/repo
mkdir -p gpurun_out/r06j
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "footprint or rccl or two_rank or collectives or disagree or real_rccl" > gpurun_out/r06j/pytest.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r06j/pytest.txt | head -5
