#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_posemb_gpu.py -q 2>&1 | tee gpurun_out/pytest_gpu31.log | tail -4
timeout 600 python tools/posemb_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phases31.log | tail -7
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","stage_ms")})'
for cfg in "3 8" "2 16"; do
  set -- $cfg
  echo "=== bench lanes=$1 chunk=$2"
  timeout 900 python bench.py --steps 192 --warmup 48 --no-cpu-baseline --lanes $1 --chunk $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run31_l$1_c$2.json | python -c "$show"
  tail -3 gpurun_out/bench.err | grep -v amdgpu.ids
done
exit 0
