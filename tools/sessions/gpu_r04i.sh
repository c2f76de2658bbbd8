#!/bin/bash
# round 4, step i: plane-fed bf16 filter gradient (k_wgrad_bf16p.hip): parity, bench, per-layer table
cd /root/repo; mkdir -p gpurun_out/r04j; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bf16_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r04j/pytest_bf16.txt
tail -12 gpurun_out/r04j/pytest_bf16.txt | cut -c1-250
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r04j/bench_bf16.json 2> gpurun_out/r04j/bench.err
cut -c1-200 gpurun_out/r04j/bench_bf16.json
EG_LIB_PATH=edgegan_amd/libedgegan_hip_exp.so EG_WGRAD_BF16P_NO_CONTIG=1 python bench.py --dtype bf16 --no-cpu-baseline --profile-steps 0 > gpurun_out/r04j/bench_bf16_nocontig.json 2>> gpurun_out/r04j/bench.err
cut -c1-160 gpurun_out/r04j/bench_bf16_nocontig.json
python tools/layer_table.py gpurun_out/r04j/layers_bf16.csv --dtype bf16 2> gpurun_out/r04j/layers.err
grep wgrad gpurun_out/r04j/layers_bf16.csv | sort -t, -k11 -n -r | head -30 | cut -c1-180
