#!/bin/bash
# Round-2 GPU call 2: GPU test tier with the rewritten sampler, sampler kernels alone under rocprofv3 (stats + the two
# PMC passes), bench at the driver's flags, producer sweep, eigensolver phases.
set -u
O=gpurun_out/r2c2
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x --tb=long 2>&1 | tail -150) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sampler -o s -- python tools/sampler_alone.py 2>&1 | tail -3) > $O/prof_sampler.log
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o f -- python tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $O/pmc_f.log
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o w -- python tools/sampler_alone.py --launches 40 2>&1 | tail -2) > $O/pmc_w.log
(timeout 100 python tools/pmc_sampler.py $O/pmc_f $O/pmc_w 961441/9938200/bsz256/hops256 $O/pmc_sampler.json 2>&1 | tail -40) > $O/pmc_summary.log
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone.csv
rm -rf $O/pmc_f $O/pmc_w $O/prof_sampler
(timeout 300 python bench.py --steps 20 --warmup 5 --allow-posemb-flags 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
for cfg in "3 4 2" "3 2 2" "3 4 3" "2 4 2" "3 1 4" "3 10 2"; do
  set -- $cfg
  (timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --allow-posemb-flags --lanes $1 --chunk $2 --depth $3 2>>$O/sweep.err | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags 2>>$O/sweep.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags --chunk 16 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk16.json
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
(timeout 300 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --allow-posemb-flags --batch-size 64 --nce-k 1024 2>$O/bench_gpus2.err | tail -1) > $O/bench_gpus2.json
cut -c1-400 $O/bench_driver.json
