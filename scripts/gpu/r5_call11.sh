#!/bin/bash
# Round 5, call 11 (second form): what does each eigensolver class cost the step?  Sustained bench with one class ablated at a time
# (timing-only variant builds: that class's rows stay zero) and with the placeholder embedding (no eigensolver at all).
set -u
O=gpurun_out/r5c11
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { n=$1; shift; (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_$n.json
python -c "
import json; d=json.loads(open('$O/bench_$n.json').read()); print('sustained $n', round(d['ms_per_step'],4), d['stage_ms'])" 2>&1 | tail -1; }
run default
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
for v in MID CHEB WAVE; do
cp gcc_amd/csrc/variants/lib_free_$v.so gcc_amd/csrc/libgcc_amd.so
run free_$v
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
run placeholder --posemb placeholder
run default_again
