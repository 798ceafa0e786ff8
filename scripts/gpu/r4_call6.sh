#!/bin/bash
# Round 4, call 6: the fused eval kernel after the LDS index clamp (flat LDS access below the aperture = memory violation),
# narrow-model tests, E2E headline tolerance, eval probe.
set -u
O=gpurun_out/r4c6
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for t in tests/test_generate_gpu.py tests/test_hidden_size_gpu.py tests/test_headline_parity_gpu.py; do
  n=$(basename $t .py)
  timeout 600 python -m pytest $t -m gpu -q --tb=short -s > $O/$n.log 2>&1
  echo "== $n: $(grep -E 'passed|failed' $O/$n.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|core dumped|fault|VIOLATION" $O/$n.log | head -8 | cut -c1-300
done
(timeout 300 python tools/eval_probe.py 2>&1 | tail -2) | tee $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -1) | tee -a $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --batch-size 64 2>&1 | tail -1) | tee -a $O/eval_probe.txt
