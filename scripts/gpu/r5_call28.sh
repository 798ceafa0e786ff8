#!/bin/bash
# Round 5, call 28: flake check of the GPU tier on the final tree: the pipeline test six times, then the whole tier twice.
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q --tb=line 2>&1 | tail -1; done
for i in 1 2; do timeout 1700 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -E "passed|failed|^/|Error" | tail -4; done
