#!/bin/bash
# Round 4, call 30: small induce class, workgroups per subgraph in the static grid (2 / 1 / 3 against the default 4)
set -u
O=gpurun_out/r4c30
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
(echo "G1 default: $(timeout 300 python tools/sampler_alone.py $G1 --time 2>&1 | grep '^hub_degree')"
echo "G2 default: $(timeout 600 python tools/sampler_alone.py $G2 --time 2>&1 | grep '^hub_degree')"
for v in g2 g1 g3; do
  L=$GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_$v.so
  echo "G1 $v: $(timeout 300 python tools/sampler_alone.py $G1 --lib $L --time 2>&1 | grep '^hub_degree')"
  echo "G2 $v: $(timeout 600 python tools/sampler_alone.py $G2 --lib $L --time 2>&1 | grep '^hub_degree')"
done) | tee $O/grid_mult.txt
