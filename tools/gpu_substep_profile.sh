#!/bin/bash
# serialised kernel trace of ONE sub-step (default 4 = the classifier's) : usage  bash tools/gpu_substep_profile.sh <k> <tag>
K=${1:-4}; TAG=${2:-ss}
cd /root/repo; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
cat > /tmp/ss_run.py <<PY
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import edgegan_amd as ea
cfg = ea.Config(batch_size=64, compute_dtype='f32x3')
net = ea.EdgeGAN(cfg, ea.init_variables(cfg, 0))
net.upload_batch(**ea.synthetic_batch(cfg))
for _ in range(2): net.train_step()
net.sync()
for _ in range(10): net.substep($K, True)
net.sync(); net.close()
PY
EG_LANES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python /tmp/ss_run.py > gpurun_out/$TAG/run.log 2>&1
DB=$(find /tmp/prof_$TAG -name '*.db' | head -1)
python tools/rocprof_summary.py $DB gpurun_out/$TAG/kernel_stats.csv 10
