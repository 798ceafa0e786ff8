#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== posemb gpu tests"
timeout 900 python -m pytest tests/test_posemb_gpu.py -m gpu -q 2>&1 | tail -4
echo "=== bench (device posemb)"
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run6.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','steps_per_sec','stage_ms','final_loss','posemb_status')})"
tail -3 gpurun_out/bench.err
echo "=== rocprof (device posemb)"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof6" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof6.err"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof6" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:44]
    print(f"{n:46s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
