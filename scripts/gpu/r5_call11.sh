#!/bin/bash
# Round 5, call 11: how much of the step does the 65..128 eigensolver class cost at all?  Sustained bench with the class ablated (timing-only
# variant build, results of that class missing), and with the two-wave class's grid cap at 128 / 512 workgroups.
set -u
O=gpurun_out/r5c11
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_$1.json
python -c "
import json; d=json.loads(open('$O/bench_$1.json').read()); print('sustained $1', round(d['ms_per_step'],4), d['stage_ms'])" 2>&1 | tail -1; }
run default
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 run pair128
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,512 run pair512
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
cp gcc_amd/csrc/variants/lib_mid_free.so gcc_amd/csrc/libgcc_amd.so
run mid_free
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
