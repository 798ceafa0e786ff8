#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the sampler kernels for both graphs from THIS build (the counter file is keyed by the source hash)
set -u
O=gpurun_out/r2pmc
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
pmc() {  # tag, counter, args
  cd /tmp && (timeout 400 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
}
pmc f1 FETCH_SIZE "--launches 40"
pmc w1 WRITE_SIZE "--launches 40"
rm -f $O/pmc_sampler.json
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1 /tmp/pmc_w1 961441/9938200/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1.log
G2="--nodes 10000000 --edges 200000000 --launches 20"
pmc f2 FETCH_SIZE "$G2"
pmc w2 WRITE_SIZE "$G2"
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) > $O/bench_with_traffic.json
python -c "
import json; d=json.loads(open('$O/bench_with_traffic.json').read()); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'])
p=json.load(open('$O/pmc_sampler.json')); print(p['source_sha256'][:12], {k:v['induce_kernel_hbm_bytes_per_launch'] for k,v in p['workloads'].items()})"
