#!/bin/bash
# Round 4, call 20: induce_kernel phase ticks, hubs vs scan-all, 16-step launches on G1 and G2
set -u
O=gpurun_out/r4c20
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
G2="--nodes 10000000 --edges 200000000"
for hd in 0 -1; do
  timeout 300 python tools/induce_phases.py --steps-per-call 16 --hub-degree $hd 2>&1 | tail -1
  timeout 600 python tools/induce_phases.py $G2 --steps-per-call 16 --hub-degree $hd 2>&1 | tail -1
done | tee $O/induce_phases.txt
