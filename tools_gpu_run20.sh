#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
for cfg in "12 2" "3 2" "2 2" "1 2"; do
  set -- $cfg
  echo "=== bench lanes=$1 depth=$2"
  timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --lanes $1 --depth $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run20_l$1.json | python -c "$show"
  tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
done
exit 0
