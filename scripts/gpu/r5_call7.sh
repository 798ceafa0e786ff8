#!/bin/bash
# Round 5, call 7 (the same programme for four builds): inverse iteration with the next rows' operands requested ahead; the same with the
# solves spread over all 16 waves; the fused tridiagonalisation (one pass, one barrier per column); the any-width weight gradients
# accumulated in fp64.  Strict eigensolver tests, the any-width GPU tests with the float64 gradient bar, phases, sustained bench.
set -u
O=gpurun_out/r5c7
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_wide_encoder_gpu.py tests/test_headline_parity_gpu.py -m gpu -q --tb=short -s > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  |hidden [0-9]+:" $O/pytest_gpu.log | head -20 | cut -c1-300
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases.txt; cut -c1-300 $O/phases.txt
for i in 1 2; do
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_sustained_$i.json
python -c "
import json; d=json.loads(open('$O/bench_sustained_$i.json').read()); print('sustained', round(d['ms_per_step'],4), d['stage_ms'])"
done
