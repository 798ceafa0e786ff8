#!/bin/bash
# Round 4, call 40: solver workgroups with a smaller LDS request (65..128 class: 132 -> 104 KB, the LU batch narrows by itself;
# block class: 143 -> 133 KB through 24 instead of 64 long-row chunk slots): sustained bench, alternating, + eigensolver phases
set -u
O=gpurun_out/r4c40
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
run() { (timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>$O/$1.err | tail -1) > $O/$1.json; python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', round(d['ms_per_step'],4), 'flags', (d.get('posemb_status') or {}).get('flags'), {k: round(v,3) for k,v in d['stage_ms'].items()})" || tail -3 $O/$1.err; }
use() { if [ $1 = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$1.so gcc_amd/csrc/libgcc_amd.so; fi; }
for v in default midlds default midlds; do use $v; run ${v}_$RANDOM; done
use midlds
(timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^total") | tee $O/phases_midlds.txt
timeout 600 python -m pytest tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -2 | tee $O/pytest_midlds.txt
use default
