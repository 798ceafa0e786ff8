#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
for cfg in "3 8 32 interleaved" "3 8 64 interleaved" "3 8 96 interleaved" "3 8 64 block" "2 16 64 interleaved"; do
  set -- $cfg
  echo "=== bench lanes=$1 chunk=$2 reserved=$3 $4"
  timeout 900 python bench.py --steps 192 --warmup 48 --no-cpu-baseline --lanes $1 --chunk $2 --reserved-cus $3 --cu-layout $4 2>gpurun_out/bench.err | tee gpurun_out/bench_run27_l$1_c$2_r$3_$4.json | python -c "$show"
  tail -3 gpurun_out/bench.err | grep -v amdgpu.ids
done
exit 0
