#!/bin/bash
# Round 4, call 19: (hub_degree, max_hubs) sweep with the adaptive threshold, back-to-back 16-step launches, G1 and G2
set -u
O=gpurun_out/r4c19
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
SW="-1:0,256:32,512:32,1024:32,2048:32,4096:32,1024:24,2048:16,4096:16,-1:0,1024:32"
G2="--nodes 10000000 --edges 200000000"
(timeout 600 python tools/sampler_alone.py --launches 30 --steps-per-call 16 --sweep=$SW 2>&1 | grep "^hub_degree") | tee $O/sweep_g1.txt
(timeout 900 python tools/sampler_alone.py $G2 --launches 12 --steps-per-call 16 --sweep=$SW 2>&1 | grep "^hub_degree") | tee $O/sweep_g2.txt
