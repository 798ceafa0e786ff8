#!/bin/bash
# Round 4, call 34: why the 20-step window of the driver's flags reads 0.97 ms and a 192-step window 0.93: window length, warm-up
# length and chunk size taken apart
set -u
O=gpurun_out/r4c34
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, args
  (timeout 400 python bench.py $2 --no-cpu-baseline 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2]', round(d['ms_per_step'],4), 'untimed', d['untimed_steps'], 'chunk', d['config']['producer_chunk'])" || tail -3 $O/$1.err
}
run a1 "--steps 20 --warmup 5"
run a2 "--steps 20 --warmup 5"
run b1 "--steps 20 --warmup 200"
run b2 "--steps 20 --warmup 400"
run c1 "--steps 40 --warmup 5"
run c2 "--steps 80 --warmup 5"
run d1 "--steps 160 --warmup 40 --chunk 10"
run d2 "--steps 160 --warmup 40"
