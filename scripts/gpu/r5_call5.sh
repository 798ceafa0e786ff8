#!/bin/bash
# Round 5, call 5: the any-width encoder / head on the device (tests/test_wide_encoder_gpu.py), train.py --hidden-size 128,
# the full-size configs[3] sampler test with the oracle-sized scratch, the entry-point tests.
set -u
O=gpurun_out/r5c5
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1500 python -m pytest tests/test_wide_encoder_gpu.py tests/test_train_main_gpu.py "tests/test_sampler_gpu.py::test_config4_full_size_bit_exact" -m gpu -q --tb=short -s > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  |hidden [0-9]+:" $O/pytest_gpu.log | head -30 | cut -c1-300
