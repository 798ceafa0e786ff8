#!/bin/bash
# Round 5, call 15: the solver classes of a call forked over two side streams (GCC_POSEMB_FORK, default on) against one in-order stream:
# strict eigensolver tests, the isolated call, the driver's window (three runs each) and the sustained run, by grid caps.
set -u
O=gpurun_out/r5c15
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_headline_parity_gpu.py tests/test_train_step_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
for f in 1 0; do (GCC_POSEMB_FORK=$f timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^total|status") > $O/phases_fork$f.txt; echo "-- fork=$f: $(tr '\n' ' ' < $O/phases_fork$f.txt | cut -c1-200)"; done
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
win fork
GCC_POSEMB_FORK=0 win nofork
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,256,64,128 win fork_cheb64_waves_half
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128,128 win fork_cheb64
GCC_POSEMB_GRID_CAPS=256,64,128,64,64,128,512,128,256 win fork_cheb128_pair256
