#!/bin/bash
# Round 5, call 16: producer lanes on a LOW-priority stream (the training step's is high already).
set -u
O=gpurun_out/r5c16
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs GCC_POSEMB_FORK=0
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range())
for p in (-1, 0, 1, 2):
    try: print(p, torch.cuda.Stream(priority=p).priority)
    except Exception as e: print(p, 'ERR', e)
"
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
win prio0
GCC_LANE_PRIORITY=1 win prio1
GCC_LANE_PRIORITY=1 GCC_POSEMB_FORK=1 win prio1_fork
win prio0_again
