#!/bin/bash
# Round 4, call 17: unscanned hub rows with a per-subgraph adaptive threshold (at most max_hubs rows): how many rows of a C2 /
# G2 ego-net are over 256 / 1024 / 4096, and the wall clock of back-to-back 16-step launches over (hub_degree, max_hubs).
set -u
O=gpurun_out/r4c17
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
[ -n "${SKIP_TESTS:-}" ] || timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest.log | head -10 | cut -c1-300
SW="-1:0,256:1,256:2,256:4,256:8,256:16,256:32,1024:2,1024:4,1024:8,1024:32,4096:4,4096:8,64:4,64:8"
G2="--nodes 10000000 --edges 200000000"
[ -n "${SKIP_TESTS:-}" ] || (timeout 300 python tools/sampler_alone.py --launches 2 --hub-stats 2>&1 | grep "^view") | tee $O/hub_stats_g1.txt
[ -n "${SKIP_TESTS:-}" ] || (timeout 600 python tools/sampler_alone.py $G2 --launches 2 --hub-stats 2>&1 | grep "^view") | tee $O/hub_stats_g2.txt
(timeout 600 python tools/sampler_alone.py --launches 30 --steps-per-call 16 --sweep=$SW 2>&1 | grep "^hub_degree") | tee $O/sweep_g1.txt
(timeout 900 python tools/sampler_alone.py $G2 --launches 12 --steps-per-call 16 --sweep=$SW 2>&1 | grep "^hub_degree") | tee $O/sweep_g2.txt
