#!/bin/bash
# Round 4, call 5: which test crashed call 4 (full logs this time), the fused eval kernel on the device, relaxed stream hand-offs.
set -u
O=gpurun_out/r4c5
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for t in tests/test_hidden_size_gpu.py tests/test_generate_gpu.py tests/test_headline_parity_gpu.py tests/test_train_step_gpu.py tests/test_overflow_regrow_gpu.py tests/test_pipeline_gpu.py; do
  n=$(basename $t .py)
  timeout 600 python -m pytest $t -m gpu -q --tb=short -s > $O/$n.log 2>&1
  echo "== $n: $(grep -E '^[0-9]+ (passed|failed)|passed|failed' $O/$n.log | tail -1)"; grep -E "^(FAILED|ERROR)|Error|core dumped|fault" $O/$n.log | head -8 | cut -c1-300
done
(timeout 300 python tools/eval_probe.py 2>&1 | tail -2) | tee $O/eval_probe.txt
(timeout 300 python tools/eval_probe.py --rw-hops 256 --nodes 100000 --edges 1000000 2>&1 | tail -1) | tee -a $O/eval_probe.txt
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -4) > $O/graph_probe.txt; cat $O/graph_probe.txt
(timeout 300 python tools/graph_probe.py --steps 200 --strict-streams 2>&1 | grep shipped) | tee -a $O/graph_probe.txt
for v in relaxed strict; do
  fl=""; [ $v = strict ] && fl="--strict-streams"
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $fl 2>$O/bench_driver_$v.err | tail -1) > $O/bench_driver_$v.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $fl 2>$O/bench_192_$v.err | tail -1) > $O/bench_192_$v.json
  for f in bench_driver_$v bench_192_$v; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); s=d['stage_rooflines']; print('$f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d.get('step_launch'), 'step_roofline', round(d['step_roofline']['frac'],4), 'fwd/bwd', round(s['gin_encoder_fwd']['ms_in_step'],3), round(s['gin_encoder_bwd']['ms_in_step'],3))" || tail -3 $O/$f.err; done
done
