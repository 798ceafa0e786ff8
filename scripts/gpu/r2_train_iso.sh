#!/bin/bash
# training stream + one sampler lane, eigensolver replaced by a placeholder, under rocprofv3 --stats (kernels undisturbed)
set -u
O=gpurun_out/${1:-r2iso}
mkdir -p $O
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_encoder_gpu.py tests/test_train_step_gpu.py -q -m gpu 2>&1 | tail -2) > $O/pytest.txt
cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_i -o i -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --lanes 1 --chunk 1 --posemb placeholder 2>/dev/null | grep '^{' | tail -1) > $GRAFT_REPO_ROOT/$O/bench_training_stream.json; cd $GRAFT_REPO_ROOT
find /tmp/prof_i -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_isolated.csv
(timeout 200 python tools/graph_probe.py 2>&1 | tail -1) > $O/graph_probe.txt
cat $O/pytest.txt $O/graph_probe.txt
