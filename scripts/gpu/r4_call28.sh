#!/bin/bash
# Round 4, call 28: the packed eval kernel (small subgraphs by windows of 64 node ids): device tests of the eval / generate path,
# tools/eval_probe.py on the three shapes of profiles/r4_eval_probe.txt
set -u
O=gpurun_out/r4c28
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_encoder_gpu.py tests/test_hidden_size_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error" $O/pytest.log | head -10 | cut -c1-300
(timeout 300 python tools/eval_probe.py --batch-size 32 2>&1 | tail -2) | tee $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py 2>&1 | tail -2) | tee -a $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -2) | tee -a $O/eval_probe.txt
