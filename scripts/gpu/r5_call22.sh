#!/bin/bash
# Round 5, call 22: the driver's window by producer chunk (default: the largest divisor of --steps up to 16 = 10 for 20 steps).
set -u
O=gpurun_out/r5c22
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
win chunk10
win chunk5 --chunk 5
win chunk5_lanes4 --chunk 5 --lanes 4
win chunk10_lanes4 --lanes 4
win chunk10_again
