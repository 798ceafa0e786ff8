#!/bin/bash
# Round-3 closing measurements, part A (G1): full GPU tier, smoke, bench at the driver's flags (with the CPU legs),
# sustained, E2E mode, eigensolver phase ticks, rocprofv3 --stats of the bench command, sampler kernels alone
# (--stats, then FETCH_SIZE and WRITE_SIZE in their own passes -> profiles/pmc_sampler.json for THIS build).
set -u
O=gpurun_out/r3fa
mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 60 2>&1 | tail -3) > $GRAFT_REPO_ROOT/$O/prof_sampler.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone.csv
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_f.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_w.log; cd $GRAFT_REPO_ROOT
rm -f $O/pmc_sampler.json
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f /tmp/pmc_w 961441/9938200/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -5) > $O/pmc_summary.log
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_default.err | tail -1) > $O/bench_default.json
(timeout 300 python bench.py --mode e2e --steps 20 --warmup 5 --cpu-seconds 8 2>$O/bench_e2e256.err | tail -1) > $O/bench_e2e256.json
(timeout 300 python bench.py --mode e2e --batch-size 32 --steps 40 --warmup 10 --no-cpu-baseline 2>$O/bench_e2e32.err | tail -1) > $O/bench_e2e32.json
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/prof_bench.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
for f in bench_driver bench_192 bench_default bench_e2e256 bench_e2e32; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d['roofline']['traffic'], round(d['roofline']['frac'],3), d.get('posemb_status',{}).get('flags'))"; done
head -4 $O/posemb_phases.txt | tail -2; tail -6 $O/posemb_phases.txt
