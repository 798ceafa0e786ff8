#!/bin/bash
# Round 3, call 16: the three train.py modes added to the GPU tier (fused E2E, adagrad, MoCo + sgd) and the full CPU-side
# consistency of the final tree on the box (cabi symbols).
set -u
O=gpurun_out/r3c16
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_train_main_gpu.py tests/test_cabi_symbols.py -q --tb=short 2>&1 | tail -30) > $O/pytest.log
grep -E "passed|failed|Error|assert" $O/pytest.log | head -20
