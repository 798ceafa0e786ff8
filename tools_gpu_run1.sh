#!/bin/bash
# one gpurun call: tests -> smoke -> bench -> rocprof (outputs under gpurun_out/)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "=== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "gfx|Compute Unit" | head -4
echo "=== pytest gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "=== bench"
timeout 600 python bench.py --steps 50 --warmup 10 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "=== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
tail -3 "$GRAFT_REPO_ROOT/gpurun_out/prof.err"
find "$GRAFT_REPO_ROOT/gpurun_out/prof" -name "*stats*" | head
