#!/bin/bash
# Round-2 closing measurements, part B: BASELINE configs[3] (the sampler alone on the 10M-node / 200M-edge graph): bench line,
# rocprofv3 --stats, FETCH_SIZE / WRITE_SIZE passes merged into the G1 PMC file; two ranks on one GPU (gloo); graph probe.
set -u
O=gpurun_out/r2fb
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python bench.py --mode sampler --steps 50 --warmup 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json
G2="--nodes 10000000 --edges 200000000"
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_g2 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/prof_g2.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_g2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_g2.csv
cd /tmp && (timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f2 -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 --launches 20 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_f.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w2 -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 --launches 20 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_w.log; cd $GRAFT_REPO_ROOT
cp profiles/pmc_sampler.json $O/pmc_sampler.json
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -12) > $O/pmc_summary.log
(timeout 300 python tools/induce_phases.py $G2 2>&1 | tail -1) > $O/induce_phases_g2.txt
(timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --batch-size 64 --nce-k 1024 2>$O/bench_gpus2.err | tail -1) > $O/bench_gpus2.json
(timeout 200 python tools/graph_probe.py 2>&1 | tail -12) > $O/graph_probe.txt
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_traffic.err | tail -1) > $O/bench_driver_with_traffic.json
cut -c1-300 $O/bench_g2_sampler.json
