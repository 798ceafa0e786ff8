#!/bin/bash
# Round-2 closing measurements, part D (after the edge-balanced gather): GPU tests, bench lines, training stream alone
set -u
O=gpurun_out/r2fd
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $O/pytest_gpu.log
tail -1 $O/pytest_gpu.log
(timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_default.err | tail -1) > $O/bench_default.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_i -o i -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --lanes 1 --chunk 1 --posemb placeholder 2>/dev/null | grep '^{' | tail -1) > $GRAFT_REPO_ROOT/$O/bench_training_stream.json; cd $GRAFT_REPO_ROOT
find /tmp/prof_i -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_isolated.csv
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_d -o d -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1) > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json; cd $GRAFT_REPO_ROOT
find /tmp/prof_d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
(timeout 200 python tools/graph_probe.py 2>&1 | tail -1) > $O/graph_probe.txt
(timeout 200 python tools/gin_phases.py 2>&1 | tail -2) > $O/gin_phases.txt
for f in bench_driver bench_default bench_192 bench_training_stream; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('produced_steps'), d.get('consumed_steps'), d['posemb_status']['flags'] if 'posemb_status' in d else None)"; done
cat $O/graph_probe.txt $O/gin_phases.txt
