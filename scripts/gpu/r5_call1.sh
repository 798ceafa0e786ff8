#!/bin/bash
# Round 5, call 1: the whole GPU tier with the new tests (G2 full-size sampler, configs[4] slices vs oracle, segmented RCCL replay),
# eigensolver phases of the new block class against its A/B knobs (GCC_POSEMB_CHEB: 7 = rounds 2-4 path, 3 = wide block only,
# 5 = filter through L2), bench lines (driver flags / sustained, default vs 7), --collectives, --mode sample-ready, --mode sampler on G2.
set -u
O=gpurun_out/r5c1
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest_gpu.log 2>&1
echo "== gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error" $O/pytest_gpu.log | head -10 | cut -c1-300
for v in 1 7 3 5; do
  (GCC_POSEMB_CHEB=$v timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases_cheb$v.txt
  echo "[cheb=$v]"; cut -c1-260 $O/phases_cheb$v.txt
done
b() {  # tag, env, flags
  (env $2 timeout 500 python bench.py $3 2>$O/bench_$1.err | tail -1) > $O/bench_$1.json
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$1.json').read()); r=d['roofline']
    print('[$1]', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'subgraphs/s | roofline', r['kernel'][:40], round(r['frac'],4), '| warmup', d['warmup'], 'prod/cons', d['produced_steps'], d['consumed_steps'], '| posemb', d.get('posemb_status'), '|', (d.get('step_launch') or '')[:40])
    c=d.get('cpu_baseline')
    if c: print('   cpu_baseline', round(c['value'] or 0), c['kind'], {k:(round(v.get('value') or 0)) for k,v in (c.get('reference_shaped') or {}).items()}, 'vs_ref_nproc', c.get('vs_reference_shaped_nproc'), 'parity', [k for k in c if k.startswith('parity')])
except Exception as e:
    print('[$1] FAILED', e); print(open('$O/bench_$1.err').read()[-1500:])
PY
}
b driver GCC_POSEMB_CHEB=1 "--steps 20 --warmup 5 --no-cpu-baseline"
b driver_old GCC_POSEMB_CHEB=7 "--steps 20 --warmup 5 --no-cpu-baseline"
b sustained GCC_POSEMB_CHEB=1 "--steps 192 --warmup 64 --no-cpu-baseline"
b sustained_old GCC_POSEMB_CHEB=7 "--steps 192 --warmup 64 --no-cpu-baseline"
b collectives GCC_POSEMB_CHEB=1 "--steps 192 --warmup 64 --no-cpu-baseline --collectives"
b sample_ready GCC_POSEMB_CHEB=1 "--mode sample-ready --steps 192 --warmup 64"
b g2_sampler GCC_POSEMB_CHEB=1 "--mode sampler --steps 64 --warmup 16"
