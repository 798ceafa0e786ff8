#!/bin/bash
# CU-masked producer streams again, now with two lanes (sustained, 192 steps): compute units kept for the training step.
set -u
O=gpurun_out/r3c35
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, args
  (timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $2 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2]', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))" | tee -a $O/summary.txt
}
run base ""
run r32 "--reserved-cus 32"
run r64 "--reserved-cus 64"
run r64b "--reserved-cus 64 --cu-layout block"
run r96 "--reserved-cus 96"
run base2 ""
