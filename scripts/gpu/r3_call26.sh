#!/bin/bash
# gin_in_kernel with its Linear weight in LDS (48.8 KiB per workgroup: no room beside a 132-KiB solver workgroup) against
# the 30-KiB variant, inside the step (sustained, lanes 2 and 3).  The library is rebuilt on the box for the second half.
set -u
O=gpurun_out/r3c26
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() { # tag, args
  (timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $2 2>$O/$1.err | tail -1) > $O/$1.json
  python -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1', '[$2]', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'))" | tee -a $O/summary.txt
}
for i in 1 2; do run ldsw1_l2_$i "--lanes 2"; run ldsw1_l3_$i "--lanes 3"; done
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) | tee -a $O/summary.txt
touch gcc_amd/csrc/encoder.hip
(make -C gcc_amd/csrc EXTRA=-DGIN_IN_LDS_W=0 2>&1 | tail -2) > $O/make.log
for i in 1 2; do run ldsw0_l2_$i "--lanes 2"; run ldsw0_l3_$i "--lanes 3"; done
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) | tee -a $O/summary.txt
