#!/bin/bash
# Round 5, call 17: gcc_posemb_set_fork by mode: the data pipeline alone (sample-ready) with 0 / 1 / 2; the training window and the
# sustained run with 2 (the block class beside the rest: one side stream).
set -u
O=gpurun_out/r5c17
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for f in 0 1 2; do
(timeout 600 python bench.py --mode sample-ready --steps 192 --warmup 64 --no-cpu-baseline --posemb-fork $f 2>$O/bench_sr.err | tail -1) > $O/bench_sample_ready_fork$f.json
python -c "
import json; d=json.loads(open('$O/bench_sample_ready_fork$f.json').read()); print('sample-ready fork=$f', round(d['ms_per_step'],4), 'ms per step', round(d['value']), d['unit'])" 2>&1 | tail -1
done
win() { n=$1; shift; for i in 1 2 3; do (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_win_${n}_$i.json; done
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline "$@" 2>$O/bench.err | tail -1) > $O/bench_sus_${n}.json
python - $O/bench_win_${n}_ $O/bench_sus_${n}.json <<'PY'
import json, sys
v = []
for i in (1, 2, 3):
    try: v.append(round(json.loads(open(sys.argv[1] + "%d.json" % i).read())["ms_per_step"], 4))
    except Exception as e: v.append(None)
try: sus = round(json.loads(open(sys.argv[2]).read())["ms_per_step"], 4)
except Exception: sus = None
print("%-28s window" % sys.argv[1].split("bench_win_")[-1], v, "sustained", sus)
PY
}
win fork0
win fork2 --posemb-fork 2
win fork0_again
