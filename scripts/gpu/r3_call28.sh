#!/bin/bash
# Grid caps, second sweep around cheb = 64 (two lanes, sustained).
set -u
O=gpurun_out/r3c28
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, caps, extra
  (GCC_POSEMB_GRID_CAPS=$2 timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline ${3:-} 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2 ${3:-}]', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))" | tee -a $O/summary.txt
}
run c64       256,64,128,64,64,64,512,128
run c48       256,64,128,64,64,48,512,128
run c32       256,64,128,64,64,32,512,128
run c40       256,64,128,64,64,40,512,128
run c64m48    256,48,128,64,64,64,512,128
run c48m48    256,48,128,64,64,48,512,128
run c64w256   256,64,128,64,64,64,256,128
run c48w256   256,64,128,64,64,48,256,64
run c64l3     256,64,128,64,64,64,512,128 "--lanes 3"
run c48l3     256,64,128,64,64,48,512,128 "--lanes 3"
run c64b      256,64,128,64,64,64,512,128
for i in 1 2; do
(GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/drv$i.err | tail -1) > $O/drv$i.json
python -c "
import json; d=json.loads(open('$O/drv$i.json').read()); print('driver flags c64 run $i', round(d['ms_per_step'],4), round(d['value']))" | tee -a $O/summary.txt
done
