#!/bin/bash
# Round 4, call 27: the sampler's device tests incl. the capped-grid test of both size classes; PMC passes + kernel stats of the
# final sampler.hip (profiles/pmc_sampler.json is keyed by the source hash); one bench line to confirm traffic is accepted.
set -u
O=gpurun_out/r4c27
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error" $O/pytest.log | head -10 | cut -c1-300
rm -f $O/pmc_sampler.json
pmc() {  # tag, counter, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
}
for S in 10 16; do
  pmc f1_$S FETCH_SIZE "--launches 24 --steps-per-call $S"
  pmc w1_$S WRITE_SIZE "--launches 24 --steps-per-call $S"
  (timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1_$S /tmp/pmc_w1_$S 961441/9938200/bsz256/hops256/steps$S $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1_$S.log
done
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
pmc f2 FETCH_SIZE "$G2"
pmc w2 WRITE_SIZE "$G2"
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f2 /tmp/pmc_w2 9964365/199372800/bsz256/hops256/steps16 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g2.log
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | tail -1) > $O/bench_driver.json
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read()); r=d['roofline']; print('bench_driver', round(d['ms_per_step'],4), round(d['value']), 'frac', round(r['frac'],3), 'traffic', r['traffic'], r['traffic_source'][:60])" || tail -3 $O/bench.err
