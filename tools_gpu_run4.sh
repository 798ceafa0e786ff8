#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee gpurun_out/smoke.log
echo "=== posemb gpu tests"
timeout 900 python -m pytest tests/test_posemb_gpu.py -m gpu -q 2>&1 | tail -5
echo "=== bench (device posemb)"
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run4.json
tail -3 gpurun_out/bench.err
echo "=== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof4" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof4.err"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof4" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
