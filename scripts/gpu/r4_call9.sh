#!/bin/bash
# Round 4, call 9: where the fused eval kernel's time goes (phase ticks), by batch size.
set -u
O=gpurun_out/r4c9
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for bs in 32 128 256; do (timeout 300 python tools/eval_probe.py --batch-size $bs 2>&1 | tail -2) | tee -a $O/eval_probe.txt; done
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -2) | tee -a $O/eval_probe.txt
(timeout 300 python -m pytest tests/test_train_step_gpu.py -m gpu -q 2>&1 | tail -2)
