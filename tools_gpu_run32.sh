#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tee gpurun_out/pytest_gpu32.log | tail -5
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
echo "=== placeholder (training stream alone)"
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --posemb placeholder --lanes 1 --chunk 1 2>/dev/null | python -c "$show"
for cfg in "3 8" "2 16"; do
  set -- $cfg
  echo "=== bench lanes=$1 chunk=$2"
  timeout 900 python bench.py --steps 192 --warmup 48 --no-cpu-baseline --lanes $1 --chunk $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run32_l$1_c$2.json | python -c "$show"
  tail -3 gpurun_out/bench.err | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof32 -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --posemb placeholder --lanes 1 --chunk 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof32.err
python - <<'PY'
import csv,sys,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof32/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} us/step {float(r['TotalDurationNs'])/1e3/60:8.1f}")
PY
exit 0
