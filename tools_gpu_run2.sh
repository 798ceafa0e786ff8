#!/bin/bash
# gpurun call 2: all gpu tests -> smoke -> bench (full step, placeholder pos-emb) -> rocprof csv
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== pytest gpu"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "=== bench"
timeout 900 python bench.py --steps 50 --warmup 10 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "=== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
tail -2 "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
find "$GRAFT_REPO_ROOT/gpurun_out/prof" -type f | head
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f"
