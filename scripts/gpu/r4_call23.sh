#!/bin/bash
# Round 4, call 23: induce classes with their own workgroup shapes (small: 256 threads x 4 per subgraph; big: 512 threads):
# device tests, per-dispatch kernel times, variants of the big class.
set -u
VARIANTS=${VARIANTS:-}
O=gpurun_out/${R4_OUT:-r4c23}
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest.log | head -10 | cut -c1-300
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 --time 2>&1 | grep "^hub_degree") > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  echo "-- $1 $(cat $O/log_st_$1.txt)"
  python - <<PY | tee $O/dispatch_$1.txt
import csv, glob, collections
f = glob.glob('/tmp/st_$1/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if 'at::' in n or 'copy' in n.lower() or 'fill' in n.lower(): continue
    k = (n, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')))
    acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k[0]:30s} grid {k[1]:>9s} wg {k[2]:>5s} calls {c:4d} avg {t / c:8.1f} us")
PY
}
G1="--launches 30 --steps-per-call 16"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
stats g1_default "$G1"
stats g2_default "$G2"
for v in $VARIANTS; do
  L=$GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_$v.so
  echo "G1 $v: $(timeout 300 python tools/sampler_alone.py $G1 --lib $L --time 2>&1 | grep '^hub_degree')"
  echo "G2 $v: $(timeout 600 python tools/sampler_alone.py $G2 --lib $L --time 2>&1 | grep '^hub_degree')"
done | tee $O/variants.txt
