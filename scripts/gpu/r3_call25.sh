#!/bin/bash
# Dense workspace classes against the sparse block class for 128 < n' (GCC_POSEMB_CHEB=0 sends them all to the dense classes):
# per-item CU-time by class; and lanes = 2 against 3 at the driver's flags.
set -u
O=gpurun_out/r3c25
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(GCC_POSEMB_CHEB=0 timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/phases_cheb0.txt
(timeout 300 python tools/posemb_phases.py 2>&1 | tail -12) > $O/phases_cheb1.txt
cat $O/phases_cheb0.txt; grep -E "cheb|total|multi" $O/phases_cheb1.txt
for i in 1 2 3; do for L in 2 3; do
(timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --lanes $L 2>$O/bench_l${L}_$i.err | tail -1) > $O/bench_l${L}_$i.json
python -c "
import json; d=json.loads(open('$O/bench_l${L}_$i.json').read()); print('driver flags lanes $L run $i', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))"
done; done
