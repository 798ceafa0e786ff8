#!/bin/bash
# Last call of the round: the full GPU tier and smoke() on the final commit, the bench line at the driver's flags, kernel stats of that command.
set -u
O=gpurun_out/r3fg
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read()); print('bench_driver', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'), d['roofline']['traffic'], round(d['roofline']['frac'],3), d['cpu_baseline']['value'])"
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
find /tmp/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
head -12 $O/kernel_stats_default.csv | cut -c1-160
