#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "=== bench (device posemb)"
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run5.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','steps_per_sec','kernel_ms','stage_ms','final_loss','posemb_status')})"
tail -3 gpurun_out/bench.err
echo "=== bench (placeholder posemb)"
timeout 600 python bench.py --steps 50 --warmup 10 --posemb placeholder --no-cpu-baseline 2>gpurun_out/bench_ph.err | tee gpurun_out/bench_run5_placeholder.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','steps_per_sec','stage_ms')})"
echo "=== rocprof (device posemb)"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof5" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof5.err"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof5" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step (35 steps):", tot/35/1e6)
for r in rows[:40]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:44]
    print(f"{n:46s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
