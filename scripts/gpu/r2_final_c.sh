#!/bin/bash
# Round-2 closing measurements, part C: smoke(), the bench command under rocprofv3 --kernel-trace --stats (all streams
# overlapping), and the training stream + one sampler lane with the eigensolver replaced by a placeholder (kernels undisturbed).
set -u
O=gpurun_out/r2fc
mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4) > $O/smoke.log
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_d -o d -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json; cd $GRAFT_REPO_ROOT
find /tmp/prof_d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_default.csv
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_i -o i -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --lanes 1 --chunk 1 --posemb placeholder 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/bench_training_stream.json; cd $GRAFT_REPO_ROOT
find /tmp/prof_i -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_isolated.csv
cat $O/smoke.log; cut -c1-200 $O/bench_training_stream.json
