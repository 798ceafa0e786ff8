#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_posemb_gpu.py -q 2>&1 | tail -3
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms","untimed_steps")})'
for cfg in "2 16" "3 8"; do
  set -- $cfg
  echo "=== bench lanes=$1 chunk=$2"
  timeout 900 python bench.py --no-cpu-baseline --lanes $1 --chunk $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run37_l$1_c$2.json | python -c "$show"
done
exit 0
