#!/bin/bash
# Round 4, call 7: fused eval kernel after the flow changes (no agent-scope fences, in-place single-tile layers, weights a
# layer ahead), block-tiled wide GIN for subgraphs over 128 nodes, wide-GIN roofline unchanged?
set -u
O=gpurun_out/r4c7
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for t in tests/test_generate_gpu.py tests/test_gin_wide_gpu.py tests/test_hidden_size_gpu.py; do
  n=$(basename $t .py)
  timeout 900 python -m pytest $t -m gpu -q --tb=short -s > $O/$n.log 2>&1
  echo "== $n: $(grep -E 'passed|failed' $O/$n.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|core dumped|fault|VIOLATION|wide GIN L" $O/$n.log | head -12 | cut -c1-300
done
(timeout 300 python tools/eval_probe.py 2>&1 | tail -1) | tee $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -1) | tee -a $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --batch-size 64 2>&1 | tail -1) | tee -a $O/eval_probe.txt
(timeout 600 python tools/gin_roofline.py 2>&1 | tail -3) | tee $O/gin_roofline.txt
