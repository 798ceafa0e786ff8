#!/bin/bash
# Round 4, call 2: headline-config parity tests (tolerances relative to the tensor's scale, fp64 truth distances), the
# overflow -> regrow -> re-sample tests, sampler tier after the ADVICE fixes.
set -u
O=gpurun_out/r4c2
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q --tb=short -s 2>&1 | tail -60) > $O/pytest_headline.log
grep -E "passed|failed|vs oracle" $O/pytest_headline.log | cut -c1-900
(timeout 900 python -m pytest tests/test_overflow_regrow_gpu.py tests/test_sampler_gpu.py tests/test_pipeline_gpu.py -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_regrow.log
tail -25 $O/pytest_regrow.log | cut -c1-300
