#!/bin/bash
# Round-3 closing measurements, part E: sampler counters / kernel stats re-collected for the final build (device_compat.h changed, so the
# hash-keyed profiles/pmc_sampler.json of part C no longer matched), then the bench lines that carry roofline.traffic.
set -u
O=gpurun_out/r3fe
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
rm -f $O/pmc_sampler.json
pmc() {  # tag, counter, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
}
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_sampler_alone_$1.csv
}
for S in 10 16; do
  pmc f1_$S FETCH_SIZE "--launches 24 --steps-per-call $S"
  pmc w1_$S WRITE_SIZE "--launches 24 --steps-per-call $S"
  (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1_$S /tmp/pmc_w1_$S 961441/9938200/bsz256/hops256/steps$S $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1_$S.log
done
stats g1_steps16 "--launches 30 --steps-per-call 16"
stats g1_steps10 "--launches 30 --steps-per-call 10"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
pmc f2 FETCH_SIZE "$G2"
pmc w2 WRITE_SIZE "$G2"
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256/steps16 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
stats g2_steps16 "$G2"
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --cpu-seconds 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json
(timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --no-cpu-baseline 2>$O/bench_g1s.err | tail -1) > $O/bench_g1_sampler.json
for f in bench_driver bench_192 bench_g2_sampler bench_g1_sampler; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); r=d['roofline']; print('$f', round(d['ms_per_step'],4), round(d['value']), 'induce frac', round(r['frac'],3), 'steps/launch', r.get('steps_per_launch'), 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'], 'e2e', d.get('stage_rooflines',{}).get('sampler_end_to_end',{}).get('frac'), d['kernel_ms_isolated'])"; done
tail -n 3 $O/bench_*.err | head -20
