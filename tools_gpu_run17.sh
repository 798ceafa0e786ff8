#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_posemb_gpu.py -x -q 2>&1 | tail -5
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","steps_per_sec","stage_ms","posemb_status")})'
for cfg in "12 2 16" "6 2 16"; do
  set -- $cfg
  echo "=== bench lanes=$1 depth=$2 hwq=$3"
  GPU_MAX_HW_QUEUES=$3 timeout 900 python bench.py --steps 150 --warmup 30 --no-cpu-baseline --lanes $1 --depth $2 2>gpurun_out/bench.err | tee gpurun_out/bench_run17_l$1_q$3.json | python -c "$show"
  tail -2 gpurun_out/bench.err | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp
echo "=== rocprof default"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof17 -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof17.err
python - <<'PY'
import csv,sys,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof17/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:48]
    print(f"{n:50s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']:>6s}")
PY
exit 0
