#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_encoder_gpu.py -q 2>&1 | tail -3
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
echo "=== placeholder (training stream alone)"
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --posemb placeholder --lanes 1 --chunk 1 2>/dev/null | python -c "$show"
echo "=== bench default"
timeout 900 python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_run41.json | python -c "$show"
exit 0
