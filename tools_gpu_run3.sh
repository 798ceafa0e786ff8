#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== pytest gpu"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
echo "=== bench (device posemb)"
timeout 900 python bench.py --steps 30 --warmup 5 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "=== bench (placeholder posemb, no cpu)"
timeout 600 python bench.py --steps 30 --warmup 5 --posemb placeholder --no-cpu-baseline 2>gpurun_out/bench_ph.err | tee gpurun_out/bench_placeholder.json
echo "=== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
tail -2 "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-60 "$f" | head -5 && python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
