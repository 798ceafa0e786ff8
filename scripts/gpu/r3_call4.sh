#!/bin/bash
# Round 3, call 4: how many CUs may the LDS-heavy solver classes (mid, sparse block) hold at once?  Bench at the driver's
# flags by grid caps "small,mid,slot,krylov,big,cheb,w48,w64" (3 producer lanes run such launches concurrently).
set -u
O=gpurun_out/r3c4
mkdir -p $O
export TMPDIR=/tmp
for caps in "256,128,128,64,64,128,512,128" "256,96,128,64,64,96,512,128" "256,64,128,64,64,64,512,128" "256,48,128,64,64,48,512,128" "256,32,128,64,64,32,512,128" "256,64,128,64,64,96,512,128" "256,96,128,64,64,64,512,128" "256,64,128,64,64,64,256,64"; do
  (GCC_POSEMB_GRID_CAPS=$caps timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms'].get('gin_fwd'), d['stage_ms'].get('gin_bwd'), [v for k,v in d['stage_ms'].items() if k.startswith('posemb')])") > $O/caps_${caps//,/_}.txt 2>&1
  echo "caps $caps: $(cat $O/caps_${caps//,/_}.txt)"
done
