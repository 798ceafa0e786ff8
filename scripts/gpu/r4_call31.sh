#!/bin/bash
# Round 4, call 31: soak of the final tree: 4096 training steps (eigensolver flags, produced == consumed, regrowth), 1920 sampler-mode
# steps on the 10M / 200M graph (983,040 subgraphs: status word clean), E2E 512 steps
set -u
O=gpurun_out/r4c31
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python bench.py --steps 4096 --warmup 64 --no-cpu-baseline 2>$O/soak.err | tail -1) > $O/bench_4096.json
(timeout 900 python bench.py --mode sampler --steps 1920 --warmup 16 --no-cpu-baseline 2>$O/soak_g2.err | tail -1) > $O/bench_g2_sampler_1920.json
(timeout 600 python bench.py --mode e2e --steps 512 --warmup 32 --no-cpu-baseline 2>$O/soak_e2e.err | tail -1) > $O/bench_e2e_512.json
for f in bench_4096 bench_g2_sampler_1920 bench_e2e_512; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), 'produced/consumed', d.get('produced_steps'), d.get('consumed_steps'), 'flags', (d.get('posemb_status') or {}).get('flags'), 'regrown', d.get('sampler_regrown'))" || tail -3 $O/*.err; done
