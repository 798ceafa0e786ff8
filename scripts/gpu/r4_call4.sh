#!/bin/bash
# Round 4, call 4: as call 3 with the ring-fed step scalars (no launch between replays) and the narrow-model tests.
# bench with and without), step_roofline + parity_step in the bench line.
set -u
O=gpurun_out/r4c4
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_overflow_regrow_gpu.py tests/test_train_step_gpu.py tests/test_pipeline_gpu.py tests/test_hidden_size_gpu.py -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest.log
grep -E "passed|failed|Error|error" $O/pytest.log | cut -c1-300 | head -30
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -4) > $O/graph_probe.txt; cat $O/graph_probe.txt
for v in graph nograph; do
  fl=""; [ $v = nograph ] && fl="--no-graph"
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $fl 2>$O/bench_driver_$v.err | tail -1) > $O/bench_driver_$v.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $fl 2>$O/bench_192_$v.err | tail -1) > $O/bench_192_$v.json
  for f in bench_driver_$v bench_192_$v; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); s=d['stage_rooflines']; print('$f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d.get('step_launch'), d.get('graph_replays_in_timed_region'), 'step_roofline', round(d['step_roofline']['frac'],4), 'fwd/bwd', round(s['gin_encoder_fwd']['ms_in_step'],3), round(s['gin_encoder_bwd']['ms_in_step'],3), 'regrown', d.get('sampler_regrown'))" || tail -3 $O/$f.err; done
done
(timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 8 2>$O/bench_full.err | tail -1) > $O/bench_full.json
python -c "
import json; d=json.loads(open('$O/bench_full.json').read()); print('full', round(d['ms_per_step'],4), json.dumps(d['cpu_baseline'].get('parity_step'))[:600])" || tail -5 $O/bench_full.err
