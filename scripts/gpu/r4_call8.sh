#!/bin/bash
# Round 4, call 8: the whole GPU tier after the mixin refactor (E2E graph replay), eval probe with launch-only timing,
# E2E bench lines with / without graph replay, the train bench at the driver's flags.
set -u
O=gpurun_out/r4c8
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1700 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest_gpu.log | head -10 | cut -c1-300
(timeout 300 python tools/eval_probe.py 2>&1 | tail -1) | tee $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -1) | tee -a $O/eval_probe.txt
for v in graph nograph; do
  fl=""; [ $v = nograph ] && fl="--no-graph"
  (timeout 400 python bench.py --mode e2e --no-cpu-baseline $fl 2>$O/bench_e2e256_$v.err | tail -1) > $O/bench_e2e256_$v.json
  (timeout 400 python bench.py --mode e2e --batch-size 32 --no-cpu-baseline $fl 2>$O/bench_e2e32_$v.err | tail -1) > $O/bench_e2e32_$v.json
  for f in bench_e2e256_$v bench_e2e32_$v; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d.get('step_launch'))" || tail -3 $O/$f.err; done
done
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
for f in bench_driver bench_192; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d.get('step_launch'))" || tail -3 $O/$f.err; done
