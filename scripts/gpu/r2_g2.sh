#!/bin/bash
# BASELINE configs[3]: the sampler alone on the 10M-node / 200M-edge graph (bench line + rocprof stats + PMC), graph built once
set -u
O=gpurun_out/r2g2
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python bench.py --mode sampler --steps 50 --warmup 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_g2 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --nodes 10000000 --edges 200000000 --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/prof_g2.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_g2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_g2.csv
cd /tmp && (timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f2 -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --nodes 10000000 --edges 200000000 --launches 20 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_f.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w2 -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --nodes 10000000 --edges 200000000 --launches 20 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_w.log; cd $GRAFT_REPO_ROOT
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256 $O/pmc_sampler_g2.json 2>&1 | tail -12) > $O/pmc_summary.log
cut -c1-400 $O/bench_g2_sampler.json
