#!/bin/bash
# Round-4 closing measurements (after the sampler rework: hub rows, size classes), same programme as r4_final_a.sh: GPU tier + smoke, sampler counters / kernel stats of this build (pmc_sampler.json is
# keyed by the source hash), the bench lines (driver's flags with the CPU leg and the parity step, sustained, E2E, sampler
# mode on both graphs), rocprofv3 --stats of the bench command, eigensolver phases, graph probe, eval probe, wide-GIN roofline.
set -u
O=gpurun_out/r4fb
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1700 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== gpu tier: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest_gpu.log | head -10 | cut -c1-300
(timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
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
(timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_b.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench_driver.err | tail -1) > $O/bench_driver_c.json
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench_192.err | tail -1) > $O/bench_192_b.json
(timeout 400 python bench.py --mode e2e --no-cpu-baseline 2>$O/bench_e2e.err | tail -1) > $O/bench_e2e256.json
(timeout 400 python bench.py --mode e2e --batch-size 32 --no-cpu-baseline 2>>$O/bench_e2e.err | tail -1) > $O/bench_e2e32.json
(timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --cpu-seconds 10 2>$O/bench_g2.err | tail -1) > $O/bench_g2_sampler.json
(timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --no-cpu-baseline 2>$O/bench_g1s.err | tail -1) > $O/bench_g1_sampler.json
(timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --no-cpu-baseline --hub-degree -1 2>$O/bench_g2_scan.err | tail -1) > $O/bench_g2_sampler_scan_all_rows.json
(timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --no-cpu-baseline --hub-degree -1 2>$O/bench_g1s_scan.err | tail -1) > $O/bench_g1_sampler_scan_all_rows.json
for f in bench_driver bench_driver_b bench_driver_c bench_192 bench_192_b bench_e2e256 bench_e2e32 bench_g2_sampler bench_g1_sampler bench_g2_sampler_scan_all_rows bench_g1_sampler_scan_all_rows; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); r=d['roofline']; print('$f', round(d['ms_per_step'],4), round(d['value']), 'induce frac', round(r['frac'],3), 'traffic', r['traffic'], 'step_roofline', (d.get('step_roofline') or {}).get('frac'), 'flags', (d.get('posemb_status') or {}).get('flags'), d.get('step_launch'))" || tail -2 $O/$f.err; done
cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_b -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
find /tmp/st_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
head -8 $O/kernel_stats_default.csv | cut -c1-170
(timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt; grep -E "multi call|total|^mid|^cheb|^wave" $O/posemb_phases.txt | cut -c1-220
