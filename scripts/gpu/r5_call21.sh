#!/bin/bash
# Round 5, call 21: the RCCL calls of the multi-GPU step recorded into the step's graph (GCC_CAPTURE_COLLECTIVES=1: one graph per ring
# slot again) against the three segments with eager hand-offs; one rank.
set -u
O=gpurun_out/r5c21
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { n=$1; shift; (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline "$@" 2>$O/bench_$n.err | tail -1) > $O/bench_$n.json
python -c "
import json; d=json.loads(open('$O/bench_$n.json').read()); print('%-22s %.4f ms per step  capture failures %s  replays %s' % ('$n', d['ms_per_step'], d.get('graph_capture_failures'), d.get('graph_replays_in_timed_region')))" 2>&1 | tail -1
grep -m3 -E "capture failed|Error" $O/bench_$n.err | cut -c1-220; }
run plain
run segments --collectives
GCC_CAPTURE_COLLECTIVES=1 run captured --collectives
GCC_CAPTURE_COLLECTIVES=1 run captured_again --collectives
run segments_again --collectives
