#!/bin/bash
# Round 5, call 13: pipeline knobs on the new defaults (two-wave 65..128 class, caps cheb 64 / one-wave teams 256, 64): heavy-phase gate,
# producer lanes, chunk size, depth.
set -u
O=gpurun_out/r5c13
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { n=$1; shift; (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_$n.json
python -c "
import json; d=json.loads(open('$O/bench_$n.json').read()); print('%-16s %.4f ms per step' % ('$n', d['ms_per_step']))" 2>&1 | tail -1; }
run default_a
GCC_POSEMB_GATE=1 run gate
run lanes3 --lanes 3
run lanes1 --lanes 1
run chunk8 --chunk 8
run chunk32 --chunk 32
run depth3 --depth 3
run reserved32 --reserved-cus 32
run default_b
