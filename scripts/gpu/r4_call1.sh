#!/bin/bash
# Round 4, first call: the new headline-config parity tests, the whole GPU tier, baseline bench + eigensolver phases of the
# round-3 build, and the two one-wave-team prototypes left unmeasured by round 3 (GCC_POSEMB_EDGE_FILL / GCC_POSEMB_EXPAND4).
set -u
O=gpurun_out/r4c1
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q --tb=short -s 2>&1 | tail -60) > $O/pytest_headline.log
grep -E "passed|failed|vs oracle" $O/pytest_headline.log | cut -c1-600
(timeout 1500 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_headline_parity_gpu.py 2>&1 | tail -30) > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log
bash scripts/gpu/standard_call.sh r4c1 phases driver bench variant:wave_protos phases
(timeout 300 python -m pytest tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -5) > $O/pytest_posemb_protos.log
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_wave_protos.so gcc_amd/csrc/libgcc_amd.so
(timeout 300 python -m pytest tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -5) > $O/pytest_posemb_protos.log
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
grep -E "passed|failed" $O/pytest_posemb_protos.log
