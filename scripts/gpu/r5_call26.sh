#!/bin/bash
# Round 5, call 26: the block class's long items first (two-ended list): strict tests, isolated call, window x3 + sustained, against the build
# before (variant built from the previous commit's posemb.hip).
set -u
O=gpurun_out/r5c26
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_headline_parity_gpu.py tests/test_pipeline_gpu.py -m gpu -q --tb=short 2>&1 | tail -2
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^cheb|^total|status") | cut -c1-200
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
print("%-14s window" % sys.argv[1].split("bench_win_")[-1], v, "sustained", sus)
PY
}
win longfirst_a
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_before_longfirst.so gcc_amd/csrc/libgcc_amd.so
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^cheb|^total") | cut -c1-200
win list_order_a
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
win longfirst_b
cp gcc_amd/csrc/variants/lib_before_longfirst.so gcc_amd/csrc/libgcc_amd.so; win list_order_b; cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
