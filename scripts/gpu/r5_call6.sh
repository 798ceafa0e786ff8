#!/bin/bash
# Round 5, call 6: eigensolver grid caps after the block class got 42 % cheaper (GCC_POSEMB_GRID_CAPS = small,mid,slot,krylov,big,cheb,w48,w64;
# default 256,64,128,64,64,96,512,128): sustained bench per setting, default first and last.
set -u
O=gpurun_out/r5c6
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() {  # tag, caps
  (GCC_POSEMB_GRID_CAPS=$2 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/err_$1.txt | tail -1) > $O/bench_$1.json
  python -c "
import json; d=json.loads(open('$O/bench_$1.json').read()); s=d['stage_ms']; print('[$1] caps $2:', round(d['ms_per_step'],4), 'ms/step | gin fwd/bwd in step', round(s['gin_fwd'],3), round(s['gin_bwd'],3), '| posemb chunk', round([v for k,v in s.items() if k.startswith('posemb')][0],2))" || tail -3 $O/err_$1.txt
}
run default_a 256,64,128,64,64,96,512,128
run cheb64 256,64,128,64,64,64,512,128
run cheb48 256,64,128,64,64,48,512,128
run mid48 256,48,128,64,64,96,512,128
run mid48_cheb64 256,48,128,64,64,64,512,128
run mid96_cheb128 256,96,128,64,64,128,512,128
run waves_half 256,64,128,64,64,96,256,64
run default_b 256,64,128,64,64,96,512,128
