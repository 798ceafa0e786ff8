#!/bin/bash
# Round 5, call 19: is the capture failure of `bench.py --collectives` in the closing run (r5final_d) reproducible?
set -u
O=gpurun_out/r5c19
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for i in 1 2 3 4; do
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --collectives 2>$O/bench_coll_$i.err | tail -1) > $O/bench_coll_$i.json
python -c "
import json; d=json.loads(open('$O/bench_coll_$i.json').read()); print('collectives run $i', round(d['ms_per_step'],4))" 2>&1 | tail -1
grep -m2 -E "Error|error" $O/bench_coll_$i.err | cut -c1-200
done
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --collectives 2>$O/bench_coll_w.err | tail -1) > $O/bench_coll_w.json
python -c "
import json; d=json.loads(open('$O/bench_coll_w.json').read()); print('collectives window', round(d['ms_per_step'],4))" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_rccl_gpu.py -m gpu -q --tb=short 2>&1 | tail -3
