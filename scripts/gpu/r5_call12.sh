#!/bin/bash
# Round 5, call 12: grid caps again, now that the 65..128 class runs on two-wave teams (GCC_POSEMB_GRID_CAPS =
# small,mid,slot,krylov,big,cheb,w48,w64,pair; default 256,64,128,64,64,96,512,128,128): sustained bench per setting.
set -u
O=gpurun_out/r5c12
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { n=$1; shift; (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_$n.json
python -c "
import json; d=json.loads(open('$O/bench_$n.json').read()); print('%-16s %.4f ms per step' % ('$n', d['ms_per_step']))" 2>&1 | tail -1; }
run default_a
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128,128 run cheb64
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,48,512,128,128 run cheb48
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,128,512,128,128 run cheb128
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128,96 run cheb64_pair96
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,256,64,128 run cheb64_waves_half
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,1024,256,128 run cheb64_waves_double
run default_b
