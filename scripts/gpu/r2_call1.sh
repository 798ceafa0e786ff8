#!/bin/bash
# Round-2 GPU call 1: full GPU test tier, bench at the driver's flags, producer-pipeline sweep, eigensolver phases,
# 2-rank launcher check.  Everything lands under gpurun_out/r2c1/.
set -u
O=gpurun_out/r2c1
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -rA 2>&1 | tail -80) > $O/pytest_gpu.log
(timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
for cfg in "3 4 2" "3 2 2" "3 4 3" "2 4 2" "3 1 4" "3 5 2"; do
  set -- $cfg
  (timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $1 --chunk $2 --depth $3 2>>$O/sweep.err | tail -1) >> $O/sweep.jsonl
done
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/sweep.err | tail -1) > $O/bench_192.json
(timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --chunk 16 2>>$O/sweep.err | tail -1) > $O/bench_192_chunk16.json
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
(timeout 300 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --batch-size 64 --nce-k 1024 2>$O/bench_gpus2.err | tail -1) > $O/bench_gpus2.json
tail -5 $O/pytest_gpu.log
cat $O/bench_driver.json | cut -c1-600
