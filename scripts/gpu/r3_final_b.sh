#!/bin/bash
# Round-3 closing measurements, part B: build() + smoke() in ONE process, BASELINE configs[3] (10M / 200M graph: bench line,
# rocprofv3 --stats, FETCH_SIZE / WRITE_SIZE for that workload), BASELINE configs[4] (wide GIN roofline, unchanged kernel).
set -u
O=gpurun_out/r3fb
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
G2="--nodes 10000000 --edges 200000000 --launches 20"
cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_g2 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/prof_g2.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_g2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_g2.csv
cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f2 -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/pmc_f2.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w2 -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $G2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/pmc_w2.log; cd $GRAFT_REPO_ROOT
cp profiles/pmc_sampler.json $O/pmc_sampler.json
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 900 python bench.py --mode sampler --steps 50 --warmup 10 --cpu-seconds 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json
python -c "
import json; d=json.loads(open('$O/bench_g2_sampler.json').read()); print('g2', round(d['ms_per_step'],4), round(d['value']), d['roofline']['traffic'], round(d['roofline']['frac'],3), d['kernel_ms_isolated'], d.get('cpu_baseline',{}).get('value'))"
(timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/gin_roofline_c5.json; cut -c1-300 $O/gin_roofline_c5.json
ls gpurun_out/ 2>/dev/null | head -3
