#!/bin/bash
# Round 4, call 29: per-kernel durations of the fused eval call (pack kernel vs general kernel vs the chain's launches)
set -u
O=gpurun_out/r4c29
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for cfg in "" "--rw-hops 256 --nodes 100000 --edges 1000000"; do
  cd /tmp && rm -rf /tmp/ev && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/ev -o s -- python $GRAFT_REPO_ROOT/tools/eval_probe.py $cfg --reps 20 2>&1 | tail -2 | cut -c1-300); cd $GRAFT_REPO_ROOT
  python - <<PY | tee -a $O/eval_kernels.txt
import csv, glob
f = glob.glob('/tmp/ev/**/*kernel_stats.csv', recursive=True)[0]
print("config: $cfg")
for r in list(csv.DictReader(open(f)))[:14]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"   {n:34s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
