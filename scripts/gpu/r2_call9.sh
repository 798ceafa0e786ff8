#!/bin/bash
# Round-2 GPU call 3: GPU tests with the Chebyshev block class, flagged-item finder, eigensolver phases, sampler rocprof + PMC,
# bench at the driver's flags, producer sweep.
set -u
O=gpurun_out/r2c9
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -120) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
(timeout 200 python tools/induce_phases.py 2>&1 | tail -3) > $O/induce_phases.txt
cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py 2>&1 | tail -3) > $GRAFT_REPO_ROOT/$O/prof_sampler.log; cd $GRAFT_REPO_ROOT
find /tmp/prof_s -name "*stats*" | head; find /tmp/prof_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone.csv
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_f.log; cd $GRAFT_REPO_ROOT
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_w.log; cd $GRAFT_REPO_ROOT
find /tmp/pmc_f | head -8 > $O/pmc_files.txt; for f in $(find /tmp/pmc_f -name '*.csv' | head -3); do echo "== $f"; head -3 $f; done >> $O/pmc_files.txt
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f /tmp/pmc_w 961441/9938200/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -40) > $O/pmc_summary.log
(timeout 300 python bench.py --steps 20 --warmup 5 --allow-posemb-flags 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
for cfg in "3 4 2" "3 10 2" "4 4 2"; do
  set -- $cfg
  (timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --allow-posemb-flags --lanes $1 --chunk $2 --depth $3 2>>$O/sweep.err | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags 2>>$O/sweep.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags --chunk 16 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk16.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags --chunk 8 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk8.json
cut -c1-300 $O/bench_driver.json
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pmc_g -o g -- python $GRAFT_REPO_ROOT/tools/gin_roofline.py --iters 2 --warmup 1 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_g.log; cd $GRAFT_REPO_ROOT
(timeout 60 python tools/pmc_gin_wide.py /tmp/pmc_g $O/pmc_gin_wide.json 2>&1 | tail -30) > $O/pmc_gin_wide.log
(timeout 200 python tools/gin_roofline.py --phases 2>&1 | tail -1) > $O/gin_roofline.json
