#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
echo "=== placeholder (training stream alone)"
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --posemb placeholder --lanes 1 --chunk 1 2>/dev/null | python -c "$show"
echo "=== bench default"
timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run38.json | python -c "$show"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof38 -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --posemb placeholder --lanes 1 --chunk 1 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof38.err
python - <<'PY'
import csv,sys,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof38/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} us/step {float(r['TotalDurationNs'])/1e3/60:8.1f}")
PY
exit 0
