#!/bin/bash
# Round 5, call 3: rotation with 16 columns in flight + two X tiles per barrier pair in the Gram phase of non-Ritz rounds;
# headline parity tests with the float64 gradient-norm bar; bench --collectives under a wrapper that prints what ends it.
set -u
O=gpurun_out/r5c3
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_posemb_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases.txt; cut -c1-300 $O/phases.txt
cat > /tmp/wrap.py <<'PY'
import sys, traceback, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["bench.py"] + sys.argv[1:]
import bench
try:
    bench.main()
    print("MAIN RETURNED", file=sys.stderr)
except BaseException as e:
    traceback.print_exc()
    print("ENDED BY", repr(e), file=sys.stderr)
PY
timeout 500 python /tmp/wrap.py --steps 64 --warmup 16 --no-cpu-baseline --collectives > $O/coll.out 2> $O/coll.err; echo "collectives rc=$?"
tail -c 600 $O/coll.out | cut -c1-600; echo; tail -25 $O/coll.err | cut -c1-300
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_sustained.json
python -c "
import json; d=json.loads(open('$O/bench_sustained.json').read()); print('sustained', round(d['ms_per_step'],4), d['stage_ms'])"
