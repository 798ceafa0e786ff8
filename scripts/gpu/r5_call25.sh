#!/bin/bash
# Round 5, call 25: four-wave 65..128 class compiled for 4 against 3 waves per SIMD: the driver's window (three runs each, twice) and sustained.
set -u
O=gpurun_out/r5c25
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 600 python -m pytest tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -2
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
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
win occ4_a
cp gcc_amd/csrc/variants/lib_quad_occ3.so gcc_amd/csrc/libgcc_amd.so; win occ3_a
cp gcc_amd/csrc/variants/lib_pair128.so gcc_amd/csrc/libgcc_amd.so; win pair128
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; win occ4_b
cp gcc_amd/csrc/variants/lib_quad_occ3.so gcc_amd/csrc/libgcc_amd.so; win occ3_b
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
