#!/bin/bash
# Round 4, call 26: 64 hub slots (variant build) against 32, back-to-back 16-step launches on G1 and G2; current default's kernel times
set -u
O=gpurun_out/r4c26
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
L=$GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_mh64.so
(echo "default build:"; timeout 300 python tools/sampler_alone.py $G1 --sweep=512:32,-1:0 2>&1 | grep '^hub_degree' | sed 's/^/G1 /'
timeout 600 python tools/sampler_alone.py $G2 --sweep=512:32,-1:0 2>&1 | grep '^hub_degree' | sed 's/^/G2 /'
echo "64-slot build:"; timeout 300 python tools/sampler_alone.py $G1 --lib $L --sweep=512:32,512:48,512:64,1024:64 2>&1 | grep '^hub_degree' | sed 's/^/G1 /'
timeout 600 python tools/sampler_alone.py $G2 --lib $L --sweep=512:32,512:48,512:64,1024:64,2048:64 2>&1 | grep '^hub_degree' | sed 's/^/G2 /') | tee $O/slots.txt
