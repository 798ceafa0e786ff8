#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","steps_per_sec","kernel_ms","stage_ms","roofline","posemb_status")})'
echo "=== gpu tests"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "=== bench placeholder posemb, 1 lane (sampler kernels undisturbed)"
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --posemb placeholder --lanes 1 --depth 1 2>gpurun_out/bench.err | tee gpurun_out/bench_run9_ph.json | python -c "$show"
echo "=== bench default"
timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run9.json | python -c "$show"
echo "=== bench 16 lanes, 24 queues"
GPU_MAX_HW_QUEUES=24 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --lanes 16 2>gpurun_out/bench.err | tee gpurun_out/bench_run9_l16.json | python -c "$show"
tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
