#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R="$GRAFT_REPO_ROOT/gpurun_out"
show='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","steps_per_sec","kernel_ms_isolated","stage_ms","roofline","posemb_status","cpu_baseline") if k in d})'
echo "=== gpu tests"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $R/pytest_gpu.log
echo "=== bench default + cpu baseline"
timeout 900 python bench.py 2>$R/bench.err | tee $R/bench_run11.json | python -c "$show"
tail -2 $R/bench.err | grep -v amdgpu.ids
stats() { python - "$1" <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:26]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:48]
    print(f"{n:50s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']:>6s}")
PY
}
cd /tmp && export TMPDIR=/tmp
echo "=== rocprof stats: default config"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof11_default -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2> $R/prof11a.err
stats $R/prof11_default
echo "=== rocprof stats: isolated (1 lane, placeholder posemb)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof11_iso -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline --lanes 1 --depth 1 --posemb placeholder > /dev/null 2> $R/prof11b.err
stats $R/prof11_iso
echo "=== pmc FETCH_SIZE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/pmc_fetch -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --lanes 1 --depth 1 --posemb placeholder > /dev/null 2> $R/pmc1.err
echo "=== pmc WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/pmc_write -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --lanes 1 --depth 1 --posemb placeholder > /dev/null 2> $R/pmc2.err
python - <<'PY'
import csv,glob,collections,os
R=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out"
for name in ("pmc_fetch","pmc_write"):
    fs=glob.glob(R+"/"+name+"/**/*counter_collection.csv", recursive=True)
    if not fs: print(name,"no counter file", glob.glob(R+"/"+name+"/**/*", recursive=True)[:5]); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k=r.get("Kernel_Name","").replace('(anonymous namespace)::','').split('(')[0]
        agg[(k,r.get("Counter_Name"))].append(float(r.get("Counter_Value",0)))
    for (k,c),v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]:
        print(f"{name} {k[:40]:42s} {c} mean {sum(v)/len(v):12.1f} n {len(v)}")
PY
