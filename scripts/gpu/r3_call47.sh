#!/bin/bash
# The 2-rank launcher on the closing build (ranks oversubscribed on the one GPU of the box: correctness of the launch path only --
# sharded seeds, key all-gather, gradient all-reduce, fused Adam + EMA + meters with grad_scale = 1 / world, status agreement).
set -u
O=gpurun_out/r3c47
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 200 python bench.py --gpus 2 --steps 10 --warmup 4 --no-cpu-baseline 2>$O/g2.err | tail -1) > $O/bench_gpus2.json
python -c "
import json; d=json.loads(open('$O/bench_gpus2.json').read()); print('bench_gpus2', d['n_gpus'], round(d['ms_per_step'],3), round(d['value']), d.get('posemb_status',{}).get('flags'), d['config'].get('parallelism'), d.get('final_loss'))"
tail -n 3 $O/g2.err
