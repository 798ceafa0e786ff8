#!/bin/bash
# Training stream after the one-round-trip prologues / fused Adam tail: parity tests of the touched kernels, launch trace, bench.
set -u
O=gpurun_out/r3c39
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_train_step_gpu.py tests/test_pipeline_gpu.py tests/test_generate_gpu.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/pytest.log
grep -E "passed|failed" $O/pytest.log
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) > $O/probe.txt
cat $O/probe.txt
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/trace_run.log; cd $GRAFT_REPO_ROOT
(python tools/stream_trace.py /tmp/tr 2>&1) > $O/stream_trace.txt
tail -n 26 $O/stream_trace.txt
(timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver1.err | tail -1) > $O/bench_driver1.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/bench_192.err | tail -1) > $O/bench_192.json
for f in bench_driver1 bench_192; do python -c "
import json; d=json.loads(open('$O/$f.json').read()); print('$f', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))"; done
