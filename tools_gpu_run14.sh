#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","steps_per_sec","stage_ms","posemb_status")})'
for cfg in "12 2 16" "6 2 16"; do
  set -- $cfg
  echo "=== bench lanes=$1 depth=$2 hwq=$3"
  GPU_MAX_HW_QUEUES=$3 timeout 900 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --lanes $1 --depth $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run14_l$1_q$3.json | python -c "$show"
  tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
done
