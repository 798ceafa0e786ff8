#!/bin/bash
# Round 5, call 18: the driver's window (--steps 20 --warmup 5; chunk 5) by producer lanes / depth / look-ahead.
set -u
O=gpurun_out/r5c18
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
win() { n=$1; shift; for i in 1 2 3; do (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_win_${n}_$i.json; done
python - $O/bench_win_${n}_ <<'PY'
import json, sys
v = []
for i in (1, 2, 3):
    try: v.append(round(json.loads(open(sys.argv[1] + "%d.json" % i).read())["ms_per_step"], 4))
    except Exception as e: v.append(None)
print("%-28s window" % sys.argv[1].split("bench_win_")[-1], v)
PY
}
win default
win depth3 --depth 3
win lanes3 --lanes 3
win lanes4 --lanes 4
win ahead1 --ahead 1
win ahead3 --ahead 3
win lanes1_depth4 --lanes 1 --depth 4
win default_again
