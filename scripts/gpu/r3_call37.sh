#!/bin/bash
# The multi-rank launcher on the final build (ranks oversubscribed on the one GPU of the box: correctness of the launch path
# only -- gloo staging, sharded seeds, key all-gather, gradient all-reduce, status agreement).
set -u
O=gpurun_out/r3c37
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline 2>$O/g2.err | tail -1) > $O/bench_gpus2.json
(timeout 900 python bench.py --gpus 8 --steps 8 --warmup 4 --no-cpu-baseline --batch-size 64 --nce-k 4096 2>$O/g8.err | tail -1) > $O/bench_gpus8.json
for f in bench_gpus2 bench_gpus8; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', d['n_gpus'], round(d['ms_per_step'],3), round(d['value']), d.get('posemb_status',{}).get('flags'), d['config'].get('parallelism'), d.get('note','')[:80])"; done
tail -n 3 $O/g2.err; tail -n 3 $O/g8.err
