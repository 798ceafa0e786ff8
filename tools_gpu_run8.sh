#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","steps_per_sec","stage_ms","final_loss","posemb_status")})'
echo "=== posemb gpu tests"
timeout 900 python -m pytest tests/test_posemb_gpu.py -m gpu -q 2>&1 | tail -3
for cfg in "8 3 8" "8 3 4" "12 2 16"; do
  set -- $cfg
  echo "=== bench lanes=$1 depth=$2 hwq=$3"
  GPU_MAX_HW_QUEUES=$3 timeout 900 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --lanes $1 --depth $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run8_l$1_q$3.json | python -c "$show"
  tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
done
echo "=== rocprof lanes=8"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof8" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --lanes 8 > /dev/null 2> "$GRAFT_REPO_ROOT/gpurun_out/prof8.err"
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof8" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:50]
    print(f"{n:52s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
