#!/bin/bash
# Round 4, call 32: the sampler's device tests incl. the new threshold / slot parametrisation
set -u
O=gpurun_out/r4c32
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error" $O/pytest.log | head -10 | cut -c1-300
