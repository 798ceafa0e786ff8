#!/bin/bash
# Round 3, call 8: cluster sweeps of the block classes by one wave; posemb GPU tier, phase ticks, bench; the 8-rank
# launcher path once (oversubscribed on one GPU: gloo staging, correctness only).
set -u
O=gpurun_out/r3c8
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_posemb_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest_posemb.log
tail -3 $O/pytest_posemb.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
cat $O/posemb_phases.txt
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench.json; cat $O/bench.json
(timeout 600 python bench.py --gpus 8 --steps 4 --warmup 2 --nodes 100000 --edges 1000000 --batch-size 32 --nce-k 1024 --rw-hops 64 --chunk 2 --no-cpu-baseline 2>$O/bench_gpus8.err | tail -1) > $O/bench_gpus8.json
cut -c1-400 $O/bench_gpus8.json; tail -n 3 $O/bench_gpus8.err
