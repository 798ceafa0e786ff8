#!/bin/bash
# Round 4, call 11: pipeline shape and eigensolver grid caps again, now that the host no longer issues the step launch by
# launch (graph replay): lanes / depth / chunk / caps at the driver's flags (20 steps) and sustained (192 steps).
set -u
O=gpurun_out/r4c11
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 300 python -m pytest tests/test_train_step_gpu.py -m gpu -q 2>&1 | tail -2)
run() {  # tag, env, flags
  (env $2 timeout 400 python bench.py --no-cpu-baseline $3 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); s=d['stage_ms']; print('$1', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'fwd/bwd', round(s['gin_fwd'],3), round(s['gin_bwd'],3), 'flags', (d.get('posemb_status') or {}).get('flags'))" 2>/dev/null || (echo "$1 FAILED"; tail -2 $O/$1.err)
}
run d_base      "X=1" "--steps 20 --warmup 5"
run s_base      "X=1" "--steps 192 --warmup 64"
run s_l3        "X=1" "--steps 192 --warmup 64 --lanes 3"
run s_d3        "X=1" "--steps 192 --warmup 64 --depth 3"
run s_c8        "X=1" "--steps 192 --warmup 64 --chunk 8"
run s_c12       "X=1" "--steps 192 --warmup 64 --chunk 12"
run s_cheb64    "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128" "--steps 192 --warmup 64"
run s_cheb128   "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,128,512,128" "--steps 192 --warmup 64"
run s_mid96     "GCC_POSEMB_GRID_CAPS=256,96,128,64,64,96,512,128" "--steps 192 --warmup 64"
run s_mid48     "GCC_POSEMB_GRID_CAPS=256,48,128,64,64,96,512,128" "--steps 192 --warmup 64"
run s_wave      "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,256,64" "--steps 192 --warmup 64"
run d_cheb64    "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128" "--steps 20 --warmup 5"
run d_c20       "X=1" "--steps 20 --warmup 5 --chunk 20"
run d_c5        "X=1" "--steps 20 --warmup 5 --chunk 5"
