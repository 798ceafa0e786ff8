#!/bin/bash
# Round 4, call 13: the N > 1 code path after the _GraphedStep refactor, on a one-GPU box (ranks share the device over gloo:
# correctness of launcher / collectives / status agreement only, NOT a scaling number), and bench.py --gpus 1 under the
# launcher against the plain run.
set -u
O=gpurun_out/${R4_OUT:-r4c13}
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --batch-size 64 --nce-k 1024 2>$O/g2.err | tail -1) > $O/bench_gpus2.json
python -c "
import json; d=json.loads(open('$O/bench_gpus2.json').read()); print('gpus2', d['n_gpus'], round(d['ms_per_step'],3), d['config']['parallelism'][:90], d.get('step_launch'), 'loss', d['final_loss'])" || tail -8 $O/g2.err
(timeout 900 python bench.py --gpus 8 --steps 8 --warmup 4 --no-cpu-baseline --batch-size 64 --nce-k 4096 2>$O/g8.err | tail -1) > $O/bench_gpus8.json
python -c "
import json; d=json.loads(open('$O/bench_gpus8.json').read()); print('gpus8', d['n_gpus'], round(d['ms_per_step'],3), d['config']['parallelism'][:90], d.get('step_launch'), 'loss', d['final_loss'])" || tail -8 $O/g8.err
(timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/l1.err | tail -1) > $O/bench_launcher1.json
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/p1.err | tail -1) > $O/bench_plain1.json
for f in bench_launcher1 bench_plain1; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', d['n_gpus'], round(d['ms_per_step'],4), round(d['value']), d.get('step_launch'))" || tail -5 $O/*.err; done
