#!/bin/bash
# Stalk deflation (pendant two-paths of a hub collapse to one): strict posemb parity tests on the device, per-class CU-time of
# the 16-view call with and without it, then the bench at the driver's flags and sustained.
set -u
O=gpurun_out/r3c19
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_pipeline_gpu.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/pytest_posemb.log
grep -E "passed|failed" $O/pytest_posemb.log
(GCC_POSEMB_STALKS=0 timeout 300 python tools/posemb_phases.py 2>&1 | tail -40) > $O/phases_stalks0.txt
(timeout 300 python tools/posemb_phases.py 2>&1 | tail -40) > $O/phases_stalks1.txt
grep -E "CU-s|call|total" $O/phases_stalks0.txt | head; grep -E "CU-s|call|total" $O/phases_stalks1.txt | head
for i in 1 2; do
(timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver$i.err | tail -1) > $O/bench_driver$i.json
done
(GCC_POSEMB_STALKS=0 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver_s0.err | tail -1) > $O/bench_driver_s0.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(GCC_POSEMB_STALKS=0 timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192_s0.err | tail -1) > $O/bench_192_s0.json
for f in bench_driver1 bench_driver2 bench_driver_s0 bench_192 bench_192_s0; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status'))"; done
