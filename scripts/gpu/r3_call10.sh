#!/bin/bash
# Round 3, call 10: producer pipeline shape after the solver work -- lanes x depth x chunk x ahead at the driver's flags.
set -u
O=gpurun_out/r3c10
mkdir -p $O
export TMPDIR=/tmp
run() { (timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms'].get('gin_fwd'),3), round(d['stage_ms'].get('gin_bwd'),3), [round(v,1) for k,v in d['stage_ms'].items() if k.startswith('posemb')])") 2>&1 | tail -1; }
for spec in "" "--lanes 4" "--lanes 2" "--depth 3" "--lanes 4 --depth 3" "--chunk 5" "--chunk 4 --lanes 4" "--ahead 4" "--ahead 5 --depth 3" "--lanes 2 --depth 3"; do
  echo "[$spec] $(run $spec)" | tee -a $O/sweep.txt
done
for caps in "256,96,128,64,64,128,512,128" "256,48,128,64,64,96,512,128" "256,64,128,64,64,128,512,128"; do
  echo "[caps $caps] $(GCC_POSEMB_GRID_CAPS=$caps run)" | tee -a $O/sweep.txt
done
