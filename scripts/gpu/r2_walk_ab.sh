#!/bin/bash
# walk kernel with all walks in flight and wave-local sort stages: sampler kernels alone on both graphs
set -u
O=gpurun_out/r2walk
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() {
  cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
}
(timeout 300 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -1) > $O/pytest.txt
run g1 "--launches 60"
run g2 "--nodes 10000000 --edges 200000000 --launches 40"
cat $O/pytest.txt
