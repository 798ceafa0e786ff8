#!/bin/bash
# Round 5, call 8: the middle induce class (subgraphs of 320 < members <= 1024: LDS tables for 1024 members, three workgroups per CU)
# against two classes (GCC_SAMPLER_MID_CLASS=0): sampler device tests incl. configs[3] at full size, kernel stats of the sampler alone
# on the 10M / 200M graph and on G1, wall clock both ways.
set -u
O=gpurun_out/r5c8
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1500 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error|^E  " $O/pytest_gpu.log | head -20 | cut -c1-300
stats() {  # tag, env, args
  rm -rf /tmp/st_$1
  (cd /tmp && env $2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 --time 2>&1 | grep -E "ms per launch|ok workload") > $O/time_$1.txt
  f=$(find /tmp/st_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$1.csv
  echo "[$1] $(head -1 $O/time_$1.txt)"; python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats_$1.csv')))
for r in rows[:7]: print('   %-60s calls %5s avg %9.1f us  %5s %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
}
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
stats g2_mid GCC_SAMPLER_MID_CLASS=1 "$G2"
stats g2_two GCC_SAMPLER_MID_CLASS=0 "$G2"
stats g1_mid GCC_SAMPLER_MID_CLASS=1 "--launches 30 --steps-per-call 16"
