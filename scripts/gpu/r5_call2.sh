#!/bin/bash
# Round 5, call 2: long rows of the block class balanced over chunk slots (threshold 32, adaptive chunk length), C-matrix tick,
# graphs of the look-ahead ring slots captured right after step 0, segmented RCCL replay test, --collectives with a fault handler.
set -u
O=gpurun_out/r5c2
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_train_step_gpu.py tests/test_posemb_gpu.py tests/test_pipeline_gpu.py tests/test_headline_parity_gpu.py tests/test_gin_wide_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
for v in 1 7; do
  (GCC_POSEMB_CHEB=$v timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases_cheb$v.txt
  echo "[cheb=$v]"; cut -c1-300 $O/phases_cheb$v.txt
done
b() {  # tag, env, flags
  (env $2 timeout 500 python bench.py $3 2>$O/bench_$1.err | tail -1) > $O/bench_$1.json; echo "[$1] rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read()); r=d['roofline']
    print('[$1]', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'subgraphs/s | roofline', r['kernel'][:40], round(r['frac'],4), '| warmup', d['warmup'], 'prod/cons', d['produced_steps'], d['consumed_steps'], '| replays', d.get('graph_replays_in_timed_region'), '|', (d.get('step_launch') or '')[:40])
except Exception as e:
    print('[$1] FAILED', e); print(open('$O/bench_$1.err').read()[-2500:])
PY
}
b driver GCC_POSEMB_CHEB=1 "--steps 20 --warmup 5 --no-cpu-baseline"
b driver2 GCC_POSEMB_CHEB=1 "--steps 20 --warmup 5 --no-cpu-baseline"
b sustained GCC_POSEMB_CHEB=1 "--steps 192 --warmup 64 --no-cpu-baseline"
b collectives GCC_POSEMB_CHEB=1 "--steps 192 --warmup 64 --no-cpu-baseline --collectives"
