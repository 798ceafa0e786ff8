#!/bin/bash
# The 20-step window of the driver's flags by producer chunk (the default takes the largest divisor of --steps not above 16 = 10).
set -u
O=gpurun_out/r3c38
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for c in 10 5 4 20 10 5; do
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --chunk $c 2>$O/c$c.err | tail -1) > $O/c$c.json
python -c "
import json; d=json.loads(open('$O/c$c.json').read()); print('chunk $c', round(d['ms_per_step'],4), round(d['value']), d.get('produced_steps'), d.get('consumed_steps'))" | tee -a $O/summary.txt
done
