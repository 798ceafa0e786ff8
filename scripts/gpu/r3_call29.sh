#!/bin/bash
# Full GPU tier + smoke on the build with stalk deflation, the trimmed training stream and two producer lanes; bench lines.
set -u
O=gpurun_out/r3c29
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --mode e2e --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_e2e.err | tail -1) > $O/bench_e2e.json
(timeout 300 python bench.py --mode e2e --batch-size 32 --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_e2e32.err | tail -1) > $O/bench_e2e32.json
for f in bench_driver bench_192 bench_e2e bench_e2e32; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'), d['roofline']['traffic'])"; done
tail -n 3 $O/*.err | head -20
