#!/bin/bash
# Round 4, call 12: soak -- 4096 timed steps (the pinned scalars ring of 2048 entries wraps twice, its run-ahead guard fires),
# losses finite, eigensolver flags 0; then train.py for 3 short epochs with checkpoint + resume.
set -u
O=gpurun_out/r4c12
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python bench.py --steps 4096 --warmup 64 --no-cpu-baseline 2>$O/bench_4096.err | tail -1) > $O/bench_4096.json
python -c "
import json; d=json.loads(open('$O/bench_4096.json').read()); print('soak', d['steps'], round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'final loss', d['final_loss'], 'flags', d['posemb_status']['flags'], 'replays', d.get('graph_replays_in_timed_region'), 'regrown', d.get('sampler_regrown'), 'produced/consumed', d['produced_steps'], d['consumed_steps'])" || tail -5 $O/bench_4096.err
(timeout 900 python -m pytest tests/test_train_main_gpu.py -m gpu -q --tb=short 2>&1 | tail -3)
