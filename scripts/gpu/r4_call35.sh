#!/bin/bash
# Round 4, call 35: the driver's flags with --steady-steps 128 (default) x3, and with 0 (the old behaviour) x1
set -u
O=gpurun_out/r4c35
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python bench.py --steps 20 --warmup 5 2>$O/r1.err | tail -1) > $O/bench_driver.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/r2.err | tail -1) > $O/bench_driver_b.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/r3.err | tail -1) > $O/bench_driver_c.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --steady-steps 0 2>$O/r4.err | tail -1) > $O/bench_driver_steady0.json
(timeout 400 python bench.py --mode e2e --no-cpu-baseline 2>$O/r5.err | tail -1) > $O/bench_e2e256.json
for f in bench_driver bench_driver_b bench_driver_c bench_driver_steady0 bench_e2e256; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), 'untimed', d['untimed_steps'], 'parity', ((d.get('cpu_baseline') or {}).get('parity_step') or {}).get('loss_rel_err'))" || tail -3 $O/*.err; done
