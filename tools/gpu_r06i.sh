#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06i
rm -f gpurun_out/r06i/probe.txt
for L in 5 15; do for C in 0 14; do
  echo "== EG_LANES=$L classes=$C" >> gpurun_out/r06i/probe.txt
  EG_LANES=$L EG_GRAPH=1 EG_GRAPH_DEBUG=1 timeout 120 python tools/gpu_graph_probe.py $C 4 6 1 >> gpurun_out/r06i/probe.txt 2>&1
  echo "rc=$?" >> gpurun_out/r06i/probe.txt
done; done
grep -E "==|rc=|end capture|instantiate|step 5|closed|HIP error" gpurun_out/r06i/probe.txt | head -80
bash tools/gpu_r06h.sh
