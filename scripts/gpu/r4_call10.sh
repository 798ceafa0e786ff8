#!/bin/bash
# Round 4, call 10: the small-subgraph eval kernel on the device (tests, probe by batch size and hop count).
set -u
O=gpurun_out/r4c10
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for t in tests/test_generate_gpu.py tests/test_hidden_size_gpu.py; do
  n=$(basename $t .py)
  timeout 600 python -m pytest $t -m gpu -q --tb=short -s > $O/$n.log 2>&1
  echo "== $n: $(grep -E 'passed|failed' $O/$n.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|core dumped|fault|VIOLATION" $O/$n.log | head -8 | cut -c1-300
done
for bs in 32 256; do (timeout 300 python tools/eval_probe.py --batch-size $bs 2>&1 | tail -2) | tee -a $O/eval_probe.txt; done
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -2) | tee -a $O/eval_probe.txt
