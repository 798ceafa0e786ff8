#!/bin/bash
# Round 4, call 14: hub rows not scanned by the induction (default threshold 256) against scanning everything (--hub-degree -1):
# device tests (bit-exact vs the C oracle up to C2 size and on the 2M / 40M graph), kernel stats of the sampler alone on G1
# and the 10M / 200M graph, sampler-mode bench lines, the train bench at the driver's flags.
set -u
O=gpurun_out/${R4C14_OUT:-r4c14}
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_overflow_regrow_gpu.py tests/test_headline_parity_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"; grep -E "^(FAILED|ERROR)|core dumped|VIOLATION" $O/pytest.log | head -10 | cut -c1-300
stats() { # tag, args
  cd /tmp && (timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_st_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/st_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
  echo "-- $1"; cut -d, -f1-4 $O/kernel_stats_$1.csv | sed 's/(anonymous namespace):://; s/(.*)",/",/' | head -10
}
stats g1_hub "--launches 30 --steps-per-call 16"
stats g1_scan "--launches 30 --steps-per-call 16 --hub-degree -1"
G2="--nodes 10000000 --edges 200000000 --launches 12 --steps-per-call 16"
stats g2_hub "$G2"
stats g2_scan "$G2 --hub-degree -1"
if [ -n "${R4C14_SWEEP:-}" ]; then   # threshold sweep: wall clock of back-to-back launches
  for hd in -1 64 128 256 512 1024 4096; do
    echo "G1 $(timeout 300 python tools/sampler_alone.py --launches 30 --steps-per-call 16 --hub-degree $hd --time 2>&1 | grep hub_degree)"
    echo "G2 $(timeout 600 python tools/sampler_alone.py $G2 --hub-degree $hd --time 2>&1 | grep hub_degree)"
  done | tee $O/hub_sweep.txt
fi
for v in hub scan; do
  fl=""; [ $v = scan ] && fl="--hub-degree -1"
  (timeout 900 python bench.py --mode sampler --steps 96 --warmup 16 --no-cpu-baseline $fl 2>$O/bench_g2_$v.err | tail -1) > $O/bench_g2_sampler_$v.json
  (timeout 300 python bench.py --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --no-cpu-baseline $fl 2>$O/bench_g1s_$v.err | tail -1) > $O/bench_g1_sampler_$v.json
  (timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $fl 2>$O/bench_driver_$v.err | tail -1) > $O/bench_driver_$v.json
  (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $fl 2>$O/bench_192_$v.err | tail -1) > $O/bench_192_$v.json
  for f in bench_g2_sampler_$v bench_g1_sampler_$v bench_driver_$v bench_192_$v; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); r=d['roofline']; print('$f', round(d['ms_per_step'],4), round(d['value']), 'induce frac', round(r['frac'],3), d['kernel_ms_isolated'])" || tail -3 $O/*_$v.err; done
done
