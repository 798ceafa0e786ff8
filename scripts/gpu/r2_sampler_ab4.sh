#!/bin/bash
# induction with per-workgroup start records and the first unit's loads issued before the bitmap build: G1, 10M/200M, phases
set -u
O=gpurun_out/r2ab4
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() {
  cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
}
(timeout 200 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -2) > $O/pytest.txt
G2="--nodes 10000000 --edges 200000000 --launches 40"
run g1 "--launches 60"
run g2 "$G2"
(timeout 300 python tools/induce_phases.py --nodes 10000000 --edges 200000000 2>&1 | tail -1) > $O/phases_g2.txt
(timeout 300 python tools/induce_phases.py 2>&1 | tail -1) > $O/phases_g1.txt
cat $O/pytest.txt $O/phases_g2.txt $O/phases_g1.txt
