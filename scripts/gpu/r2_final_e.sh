#!/bin/bash
# Round-2 closing run of the final build: the whole GPU tier and the bench line at the driver's flags
set -u
O=gpurun_out/r2fe
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
(timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
for f in bench_driver bench_192; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d['produced_steps'], d['consumed_steps'], d['posemb_status']['flags'], d['roofline']['frac'], d['roofline']['traffic'])"; done
