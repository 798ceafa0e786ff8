#!/bin/bash
# Round-4 last check of the final tree: the whole GPU tier + smoke, and the headline line once more
set -u
O=gpurun_out/r4fc
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1700 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest_gpu.log | head -10 | cut -c1-300
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1) > $O/smoke.log; cat $O/smoke.log
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_driver.json
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read()); r=d['roofline']; print('bench_driver', round(d['ms_per_step'],4), round(d['value']), 'frac', round(r['frac'],3), 'traffic', r['traffic'])" || tail -3 $O/bench.err
