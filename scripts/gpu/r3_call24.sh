#!/bin/bash
# Who slows the training stream (0.716 ms alone) down inside the step now?  Sustained runs (192 steps) with the sampler
# lanes only (--posemb placeholder), with the eigensolver, and by lanes / chunk.
set -u
O=gpurun_out/r3c24
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, args
  (timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $2 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2]', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))" | tee -a $O/summary.txt
}
run placeholder "--posemb placeholder"
run device ""
run lanes2 "--lanes 2"
run lanes4 "--lanes 4"
run lanes1 "--lanes 1"
run chunk8 "--chunk 8"
run chunk12 "--chunk 12"
run depth3 "--depth 3"
run lanes2_depth3 "--lanes 2 --depth 3"
