#!/bin/bash
# Round 5, call 14: the DRIVER's window (--steps 20 --warmup 5) by eigensolver grid caps and 65..128 class (three runs each): the short
# window right after start-up weighs a call's latency, the sustained run its CU-time.
set -u
O=gpurun_out/r5c14
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { n=$1; shift; for i in 1 2 3; do (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_${n}_$i.json; done
python - $O/bench_${n}_ <<'PY'
import json, sys
v = []
for i in (1, 2, 3):
    try: v.append(round(json.loads(open(sys.argv[1] + "%d.json" % i).read())["ms_per_step"], 4))
    except Exception as e: v.append(None)
print("%-28s" % sys.argv[1].split("bench_")[-1], v)
PY
}
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1     # (first process of the box: page-in)
run new_caps
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 run old_caps_pair128
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,256 run old_caps_pair256
GCC_POSEMB_PAIR=0 GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,128 run old_caps_1024_thread_class
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,256,64,128 run cheb96_waves_half
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,128,512,128,256 run cheb128_pair256
