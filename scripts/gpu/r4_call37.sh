#!/bin/bash
# Round 4, call 37: does LDS co-residency beside the solver workgroups matter?  The 65..128 class requests 132 KB (28 KB left: one
# 26 KB training workgroup fits beside it); a variant requests 12 KB more (none fits).  Sustained bench with both libraries, twice.
set -u
O=gpurun_out/r4c37
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
run() { (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/$1.err | tail -1) > $O/$1.json; python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', round(d['ms_per_step'],4), d['stage_ms'])" || tail -3 $O/$1.err; }
run default_1
cp gcc_amd/csrc/variants/lib_midpad.so gcc_amd/csrc/libgcc_amd.so; run midpad_1
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; run default_2
cp gcc_amd/csrc/variants/lib_midpad.so gcc_amd/csrc/libgcc_amd.so; run midpad_2
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
