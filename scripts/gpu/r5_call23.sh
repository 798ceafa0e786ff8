#!/bin/bash
# Round 5, call 23: the driver's window by the two-wave class's grid cap between 128 and 256 (three runs each).
set -u
O=gpurun_out/r5c23
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
win() { n=$1; shift; for i in 1 2 3; do (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_win_${n}_$i.json; done
python - $O/bench_win_${n}_ <<'PY'
import json, sys
v = []
for i in (1, 2, 3):
    try: v.append(round(json.loads(open(sys.argv[1] + "%d.json" % i).read())["ms_per_step"], 4))
    except Exception as e: v.append(None)
print("%-18s window" % sys.argv[1].split("bench_win_")[-1], v)
PY
}
win pair128
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,160 win pair160
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,192 win pair192
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,96 win pair96
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,80,512,128,128 win cheb80
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,112,512,128,128 win cheb112
win pair128_again
