#!/bin/bash
# Round 3, call 7: does a CU partition (producer streams masked off N compute units) beat grid caps now that the small
# solver classes are light?  Bench at 40 steps by --reserved-cus and caps.
set -u
O=gpurun_out/r3c7
mkdir -p $O
export TMPDIR=/tmp
run() { # caps reserved layout
  (GCC_POSEMB_GRID_CAPS=$1 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --reserved-cus $2 --cu-layout $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms'].get('gin_fwd'),3), round(d['stage_ms'].get('gin_bwd'),3), [round(v,1) for k,v in d['stage_ms'].items() if k.startswith('posemb')])") 2>&1 | tail -1
}
for spec in "256,64,128,64,64,96,512,128 0 interleaved" "256,128,128,64,64,128,512,128 64 interleaved" "256,192,128,64,64,192,768,192 64 interleaved" "256,192,128,64,64,192,768,192 96 interleaved" "256,192,128,64,64,192,768,192 64 block" "256,128,128,64,64,128,512,128 32 interleaved" "256,96,128,64,64,128,512,128 0 interleaved"; do
  set -- $spec
  echo "caps $1 reserved $2 $3: $(run $1 $2 $3)" | tee -a $O/sweep.txt
done
