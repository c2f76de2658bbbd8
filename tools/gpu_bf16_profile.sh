export TMPDIR=/tmp
mkdir -p gpurun_out/bf1
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf -o trace -- python bench.py --dtype bf16 --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > gpurun_out/bf1/bench.json 2> gpurun_out/bf1/err.txt
DB=$(find /tmp/prof_bf -name '*.db' | head -1)
python tools/rocprof_summary.py $DB gpurun_out/bf1/kernel_stats.csv 13
