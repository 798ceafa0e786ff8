#!/bin/bash
# Round 4, call 41: the block class on 110 KB of LDS (dense-phase matrices overlay the edge list; 143 KB before: variant lib_oldcheb.so):
# strict device tests, eigensolver phases, sustained bench alternating old / new
set -u
O=gpurun_out/r4c41
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 600 python -m pytest tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -3 | tee $O/pytest.txt
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^slot|^total") | tee $O/phases_new.txt
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_new.so
run() { (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/$1.err | tail -1) > $O/$1.json; python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', round(d['ms_per_step'],4), 'flags', (d.get('posemb_status') or {}).get('flags'), {k: round(v,3) for k,v in d['stage_ms'].items()})" || tail -3 $O/$1.err; }
use() { if [ $1 = new ]; then cp /tmp/lib_new.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$1.so gcc_amd/csrc/libgcc_amd.so; fi; }
for v in new oldcheb new oldcheb; do use $v; run ${v}_$RANDOM; done
use oldcheb
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^cheb|^slot|^total") | tee $O/phases_old.txt
use new
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/drv.err | tail -1) > $O/bench_driver.json; python -c "
import json; d=json.loads(open('$O/bench_driver.json').read()); print('driver flags (new)', round(d['ms_per_step'],4))"
