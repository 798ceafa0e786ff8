#!/bin/bash
# Round 4, call 21: walk / induce split into size classes (LDS per class instead of per graph maximum): device tests, kernel
# stats hubs vs scan-all on G1 and G2, back-to-back wall clock.
set -u
O=gpurun_out/${R4_OUT:-r4c21}
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest.log | head -10 | cut -c1-300
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 --time 2>&1 | grep "^hub_degree") > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
  echo "-- $1 $(cat $O/log_st_$1.txt)"
  python - <<PY
import csv
for r in csv.DictReader(open('$O/kernel_stats_$1.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'at::' in n or 'copy' in n or 'fill' in n.lower(): continue
    print(f"   {n:34s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total/launch {float(r['TotalDurationNs'])/1e3/$3:9.1f}")
PY
}
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
stats g1_hub "$G1" 60
stats g1_scan "$G1 --hub-degree -1" 60
stats g2_hub "$G2" 24
stats g2_scan "$G2 --hub-degree -1" 24
