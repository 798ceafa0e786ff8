#!/bin/bash
# Round 4, call 36: 20 steps per sampler call / 40 views per eigensolver call (GCC_SAMPLE_MAX_STEPS 16 -> 20, GCC_POSEMB_MAX_VIEWS
# 32 -> 40; bench.py --chunk 20): device tests, PMC passes for the 20-step launch shape, the headline line and a sustained one
set -u
O=gpurun_out/r4c36
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 1200 python -m pytest tests/test_posemb_gpu.py tests/test_pipeline_gpu.py tests/test_sampler_gpu.py tests/test_train_step_gpu.py tests/test_overflow_regrow_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION|Error" $O/pytest.log | head -10 | cut -c1-300
cp profiles/pmc_sampler.json $O/pmc_sampler.json
pmc() {  # tag, counter, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --pmc $2 --kernel-trace -d /tmp/pmc_$1 -o p -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $3 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
}
pmc f1_20 FETCH_SIZE "--launches 24 --steps-per-call 20"
pmc w1_20 WRITE_SIZE "--launches 24 --steps-per-call 20"
(timeout 100 python tools/pmc_sampler.py /tmp/pmc_f1_20 /tmp/pmc_w1_20 961441/9938200/bsz256/hops256/steps20 $O/pmc_sampler.json 2>&1 | tail -3) > $O/summary_g1_20.log
cp $O/pmc_sampler.json profiles/pmc_sampler.json
(timeout 600 python bench.py --steps 20 --warmup 5 2>$O/r1.err | tail -1) > $O/bench_driver.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/r2.err | tail -1) > $O/bench_driver_b.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/r3.err | tail -1) > $O/bench_driver_c.json
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --chunk 10 2>$O/r4.err | tail -1) > $O/bench_driver_chunk10.json
(timeout 400 python bench.py --steps 200 --warmup 40 --no-cpu-baseline 2>$O/r5.err | tail -1) > $O/bench_200.json
(timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/r6.err | tail -1) > $O/bench_192.json
for f in bench_driver bench_driver_b bench_driver_c bench_driver_chunk10 bench_200 bench_192; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); r=d['roofline']; print('$f', round(d['ms_per_step'],4), round(d['value']), 'chunk', d['config']['producer_chunk'], 'untimed', d['untimed_steps'], 'traffic', r['traffic'], 'flags', (d.get('posemb_status') or {}).get('flags'))" || tail -3 $O/*.err; done
