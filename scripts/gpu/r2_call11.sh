#!/bin/bash
set -u
O=gpurun_out/r2c11
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40) > $O/pytest_gpu.log
tail -2 $O/pytest_gpu.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -9) > $O/posemb_phases.txt
(timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
(timeout 300 python bench.py --no-cpu-baseline 2>>$O/sweep.err | tail -1) > $O/bench_default.json
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --chunk 4 2>>$O/sweep.err | tail -1) > $O/bench_20_chunk4.json
(timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --batch-size 64 --nce-k 1024 2>$O/bench_gpus2.err | tail -1) > $O/bench_gpus2.json
cut -c1-250 $O/bench_driver.json
