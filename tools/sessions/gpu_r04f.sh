#!/bin/bash
# round 4, step f: producer-written bf16 planes (prelu / gate_prelu / pool2_bwd / minmax_bwd / InstanceNorm fwd+bwd): bf16 test
# file, bench A/B against on-demand conversion (EG_NO_PRODUCER_PLANES, experiments build), serialised kernel trace
cd /root/repo; mkdir -p gpurun_out/r04f; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bf16_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r04f/pytest_bf16.txt
tail -4 gpurun_out/r04f/pytest_bf16.txt | cut -c1-250
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04f/bench_bf16.json 2> gpurun_out/r04f/bench.err
cut -c1-200 gpurun_out/r04f/bench_bf16.json
EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 > gpurun_out/r04f/bench_bf16_exp.json 2>> gpurun_out/r04f/bench.err
EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so EG_NO_PRODUCER_PLANES=1 python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 > gpurun_out/r04f/bench_bf16_noprod.json 2>> gpurun_out/r04f/bench.err
EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so EG_NO_BF16P=1 python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 > gpurun_out/r04f/bench_bf16_nobf16p.json 2>> gpurun_out/r04f/bench.err
cut -c1-160 gpurun_out/r04f/bench_bf16_exp.json gpurun_out/r04f/bench_bf16_noprod.json gpurun_out/r04f/bench_bf16_nobf16p.json
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -o trace -- python bench.py --dtype bf16 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > gpurun_out/r04f/kernel_stats_bf16.bench.json 2> gpurun_out/r04f/prof.err
DB=$(find /tmp/prof_bf -name '*.db' | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r04f/kernel_stats_bf16.csv 13
head -25 gpurun_out/r04f/kernel_stats_bf16.csv | cut -c1-150
