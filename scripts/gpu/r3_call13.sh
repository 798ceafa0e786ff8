#!/bin/bash
# Round 3, call 13: who slows the training stream down?  Training stage times with (a) no producer work in the timed
# window at all (huge look-ahead is not possible; instead: placeholder positional embedding = sampler launches only),
# (b) the full pipeline, (c) the full pipeline with one lane.
set -u
O=gpurun_out/r3c13
mkdir -p $O
export TMPDIR=/tmp
run() { (timeout 300 python bench.py --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(round(d['ms_per_step'],4), 'fwd', round(s.get('gin_fwd'),3), 'bwd', round(s.get('gin_bwd'),3), 'nce', round(s.get('nce_fwd')+s.get('nce_bwd'),3), 'sampler', round(s.get('sampler',0),3), [round(v,1) for k,v in s.items() if k.startswith('posemb')])") 2>&1 | tail -1; }
echo "[placeholder posemb, 3 lanes, 160 steps] $(run --steps 160 --posemb placeholder)" | tee -a $O/sweep.txt
echo "[placeholder posemb, 1 lane, 160 steps] $(run --steps 160 --posemb placeholder --lanes 1)" | tee -a $O/sweep.txt
echo "[full, 3 lanes, 160 steps] $(run --steps 160)" | tee -a $O/sweep.txt
echo "[full, 1 lane, 160 steps] $(run --steps 160 --lanes 1)" | tee -a $O/sweep.txt
echo "[full, 3 lanes, wave caps 256/64] $(GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,256,64 run --steps 160)" | tee -a $O/sweep.txt
echo "[full, 3 lanes, wave caps 128/32] $(GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,32 run --steps 160)" | tee -a $O/sweep.txt
(timeout 200 python tools/graph_probe.py 2>&1 | tail -4) | tee $O/graph_probe.txt
