#!/bin/bash
# Round 5, call 24: the 65..128 class on FOUR waves (thread = half a row, 64 registers) compiled for 4 / 3 / 2 waves per SIMD against the
# two-wave version (variant builds): strict eigensolver tests on the default, phases and sustained bench of each.
set -u
O=gpurun_out/r5c24
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_headline_parity_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
one() { n=$1
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^total|status") > $O/phases_$n.txt; echo "-- $n"; cut -c1-300 $O/phases_$n.txt
for i in 1 2; do (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_${n}_$i.json; done
python -c "
import json
print('sustained $n', [round(json.loads(open('$O/bench_${n}_%d.json' % i).read())['ms_per_step'],4) for i in (1,2)])" 2>&1 | tail -1; }
one quad_occ4
for v in quad_occ3 quad_occ2 pair128; do cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; one $v; done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
one quad_occ4_again
