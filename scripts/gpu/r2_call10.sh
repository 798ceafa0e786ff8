#!/bin/bash
set -u
O=gpurun_out/r2c10
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -9) > $O/posemb_phases.txt
for cfg in "3 4 2" "3 8 2" "3 10 2" "3 16 2"; do
  set -- $cfg
  (timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --allow-posemb-flags --lanes $1 --chunk $2 --depth $3 2>>$O/sweep.err | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags 2>>$O/sweep.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags --chunk 8 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk8.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --allow-posemb-flags --chunk 16 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk16.json
