#!/bin/bash
# Round-2 closing measurements, part A (G1): GPU tests, bench at the driver's flags and at the defaults, eigensolver and
# induction phase ticks, sampler kernels alone under rocprofv3 (--stats, then FETCH_SIZE and WRITE_SIZE in their own passes).
set -u
O=gpurun_out/r2fa
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 60 2>&1 | tail -3) > $GRAFT_REPO_ROOT/$O/prof_sampler.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone.csv
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_f.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_w.log; cd $GRAFT_REPO_ROOT
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f /tmp/pmc_w 961441/9938200/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -40) > $O/pmc_summary.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -9) > $O/posemb_phases.txt
(timeout 200 python tools/induce_phases.py 2>&1 | tail -1) > $O/induce_phases.txt
(timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_default.err | tail -1) > $O/bench_default.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
cut -c1-250 $O/bench_driver.json
