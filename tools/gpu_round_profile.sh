#!/bin/bash
# Collects the per-round artefacts kept under profiles/: PMC summaries (MFMA utilisation, HBM traffic), bench lines,
# serialised rocprofv3 kernel trace, per-layer table.   usage (on the GPU box): bash tools/gpu_round_profile.sh <tag>
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# PMC passes first (counters only, one group per run): the bench line's roofline.traffic reads the derived table
EG_LANES=0 bash tools/gpu_pmc.sh $OUT/pmc "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" -- python bench.py --no-side-legs --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/pmc.log 2>&1
python tools/pmc_derive.py $OUT/pmc/summary.csv $OUT/pmc_utilisation.csv && cp $OUT/pmc_utilisation.csv profiles/pmc_utilisation_latest.csv
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
python bench.py --no-side-legs --mode loop --steps 30 --no-cpu-baseline > $OUT/bench_loop.json 2>> $OUT/bench.err
python bench.py --no-side-legs --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>> $OUT/bench.err
python bench.py --no-side-legs --mode infer --no-cpu-baseline > $OUT/bench_infer.json 2>> $OUT/bench.err
python bench.py --no-side-legs --height 256 --width 512 --batch 16 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_256x512_b16.json 2>> $OUT/bench.err
python bench.py --no-side-legs --height 256 --width 512 --batch 16 --dtype bf16 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_256x512_b16_bf16.json 2>> $OUT/bench.err
python tools/layer_table.py $OUT/layers.csv 2> $OUT/layers.err
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python bench.py --no-side-legs --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-alt > $OUT/kernel_stats.bench.json 2> $OUT/prof.err
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB $OUT/kernel_stats.csv 13; fi
ls -la $OUT
