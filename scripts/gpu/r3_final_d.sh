#!/bin/bash
# Closing measurements of round 3: full GPU tier, smoke, the bench lines (driver's flags, sustained, E2E at bsz 256 / 32),
# rocprofv3 --stats of the bench command, eigensolver phases, the training stream alone.
set -u
O=gpurun_out/r3fd
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_b.json
(timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_c.json
(timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_b.json
(timeout 300 python bench.py --mode e2e --no-cpu-baseline 2>$O/bench_e2e.err | tail -1) > $O/bench_e2e256.json
(timeout 300 python bench.py --mode e2e --batch-size 32 --no-cpu-baseline 2>>$O/bench_e2e.err | tail -1) > $O/bench_e2e32.json
for f in bench_driver bench_driver_b bench_driver_c bench_192 bench_192_b bench_e2e256 bench_e2e32; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'), d['roofline']['traffic'], round(d['roofline']['frac'],3))"; done
(timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt; grep -E "multi call|total" $O/posemb_phases.txt
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) | tee $O/graph_probe.txt
cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_b -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
find /tmp/st_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
head -6 $O/kernel_stats_default.csv | cut -c1-160
tail -n 2 $O/*.err | head -20
