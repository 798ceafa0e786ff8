#!/bin/bash
# Round 5, call 10: the 65..128 eigensolver class on two-wave teams (matrix rows in registers, four workgroups per CU) against the
# 1,024-thread LDS-resident version (GCC_POSEMB_PAIR=0): strict eigensolver tests, phases of both, sustained bench of both.
set -u
O=gpurun_out/r5c10
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_headline_parity_gpu.py -m gpu -q --tb=short -s > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
for pair in 1 0; do
(GCC_POSEMB_PAIR=$pair timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases_pair$pair.txt; echo "-- pair=$pair"; cut -c1-300 $O/phases_pair$pair.txt
done
run() { (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_$1.json
python -c "
import json; d=json.loads(open('$O/bench_$1.json').read()); print('sustained $1', round(d['ms_per_step'],4), d['stage_ms'])" 2>&1 | tail -1; }
run pair256
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 run pair128
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,64 run pair64
GCC_POSEMB_PAIR=0 run pair_off
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 run pair128_again
