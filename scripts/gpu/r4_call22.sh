#!/bin/bash
# Round 4, call 22: induce workgroup shape variants after the class split (threads, grid multiplier), per-dispatch split of the
# two induce launches.
set -u
O=gpurun_out/r4c22
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
stats() { # tag, args, launches
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 --time 2>&1 | grep "^hub_degree") > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  echo "-- $1 $(cat $O/log_st_$1.txt)"
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/st_$1/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'at::' in n or 'copy' in n.lower() or 'fill' in n.lower(): continue
    k = (n, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('LDS_Block_Size', '?'))
    acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k[0]:30s} grid {k[1]:>9s} lds {k[2]:>7s} calls {c:4d} avg {t / c:8.1f} us")
PY
}
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
stats g1_default "$G1"
stats g2_default "$G2"
for v in t256 t256g4 g4 t256v16; do
  L=$GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_$v.so
  echo "G1 $v: $(timeout 300 python tools/sampler_alone.py $G1 --lib $L --time 2>&1 | grep '^hub_degree')"
  echo "G2 $v: $(timeout 600 python tools/sampler_alone.py $G2 --lib $L --time 2>&1 | grep '^hub_degree')"
done | tee $O/variants.txt
