#!/bin/bash
# Round 4, call 18: unscanned hub rows, adaptive threshold, 32 slots: kernel stats (default = 4-ary pair search, the binary-search
# variant build, scan everything) on G1 and G2 + back-to-back wall clock.
set -u
O=gpurun_out/r4c18
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 --time 2>&1 | grep "^hub_degree") > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
  echo "-- $1 $(cat $O/log_st_$1.txt)"
  python - <<PY
import csv
for r in csv.DictReader(open('$O/kernel_stats_$1.csv')):
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if 'at::' in n or 'copy' in n: continue
    print(f"   {n:20s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
}
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
V=gcc_amd/csrc/variants/lib_pairbin.so
stats g1_hub "$G1"
stats g1_pairbin "$G1 --lib $GRAFT_REPO_ROOT/$V"
stats g1_scan "$G1 --hub-degree -1"
stats g2_hub "$G2"
stats g2_pairbin "$G2 --lib $GRAFT_REPO_ROOT/$V"
stats g2_scan "$G2 --hub-degree -1"
