#!/bin/bash
# Round 5, call 9: per-dispatch durations of the induce launches on the 10M / 200M graph with and without the middle class, by grid
set -u
O=gpurun_out/r5c9
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py -m gpu -q --tb=short -x > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -10 | cut -c1-300
trace() {  # tag, env
  rm -rf /tmp/tr_$1
  (cd /tmp && env $2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$1 -o t -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --nodes 10000000 --edges 200000000 --launches 8 --steps-per-call 16 --time 2>&1 | grep -E "ms per launch") > $O/time_$1.txt
  f=$(find /tmp/tr_$1 -name "*kernel_trace.csv" | head -1)
  echo "[$1] $(cat $O/time_$1.txt)"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "induce_kernel" not in n: continue
    key = ("induce<512>" if "512" in n.split("(")[0] else "induce<256>", r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("LDS_Block_Size") or r.get("LDS_Block_Size_In_Bytes"))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]      # the timed half
    print("   ", k, "dispatches", len(v), "avg %.1f us" % (sum(v) / len(v)))
PY
}
trace mid768 "GCC_SAMPLER_MID_CLASS=1"
trace mid512 "GCC_SAMPLER_MID_CLASS=1 GCC_SAMPLER_MID_GRID=512"
trace mid256 "GCC_SAMPLER_MID_CLASS=1 GCC_SAMPLER_MID_GRID=256"
trace two "GCC_SAMPLER_MID_CLASS=0"
