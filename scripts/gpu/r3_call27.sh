#!/bin/bash
# Grid caps of the solver classes with two producer lanes (sustained, 192 steps): small,mid,slot,krylov,big,cheb,w48,w64
set -u
O=gpurun_out/r3c27
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, caps
  (GCC_POSEMB_GRID_CAPS=$2 timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2]', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))" | tee -a $O/summary.txt
}
run base      256,64,128,64,64,96,512,128
run mid96     256,96,128,64,64,96,512,128
run mid128    256,128,128,64,64,96,512,128
run mid48     256,48,128,64,64,96,512,128
run cheb64    256,64,128,64,64,64,512,128
run cheb128   256,64,128,64,64,128,512,128
run cheb160   256,64,128,64,64,160,512,128
run w48_256   256,64,128,64,64,96,256,128
run w48_1024  256,64,128,64,64,96,1024,256
run m96c128   256,96,128,64,64,128,512,128
run base2     256,64,128,64,64,96,512,128
