#!/bin/bash
# Eigensolver with the ego-net's CSR staged in LDS: parity tests, phases of a 16-view call, bench.
set -u
O=gpurun_out/r3c48
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python -m pytest tests/test_posemb_gpu.py tests/test_pipeline_gpu.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/pytest.log
grep -E "passed|failed" $O/pytest.log
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases.txt
cut -c1-250 $O/phases.txt
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
for f in bench_192 bench_driver; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))"; done
