#!/bin/bash
# Round 3, call 11: SUSTAINED throughput (160 timed steps = 10 chunks of 16) by grid caps and lanes; the 20-step window
# of the driver's flags only sees two producer launches in the background.
set -u
O=gpurun_out/r3c11
mkdir -p $O
export TMPDIR=/tmp
run() { (timeout 300 python bench.py --steps 160 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms'].get('gin_fwd'),3), round(d['stage_ms'].get('gin_bwd'),3), [round(v,1) for k,v in d['stage_ms'].items() if k.startswith('posemb')])") 2>&1 | tail -1; }
for caps in "256,64,128,64,64,96,512,128" "256,96,128,64,64,128,512,128" "256,128,128,64,64,128,512,128" "256,96,128,64,64,96,512,128" "256,80,128,64,64,112,512,128"; do
  echo "[caps $caps] $(GCC_POSEMB_GRID_CAPS=$caps run)" | tee -a $O/sweep.txt
done
for spec in "--lanes 2" "--lanes 4" "--chunk 10" "--chunk 8"; do
  echo "[caps 96/128 $spec] $(GCC_POSEMB_GRID_CAPS=256,96,128,64,64,128,512,128 run $spec)" | tee -a $O/sweep.txt
done
