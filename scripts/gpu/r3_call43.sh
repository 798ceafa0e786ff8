#!/bin/bash
# Footprint of gin_in / gin_bwd_c under contention: the bench with smaller-footprint builds swapped in (on the box only).
# v1: gin_in without the LDS weight, 4 gathered rows in flight, 128 registers (4 workgroups per CU); v3: v1 + gin_bwd_c at 128 registers;
# v4: 128 registers for both, LDS weight kept.
set -u
O=gpurun_out/r3c43
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
for v in default v1 v3 v4 default; do
  if [ $v = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; fi
  (timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline 2>>$O/bench.err | tail -1) > $O/bench_$v.json
  python -c "
import json; d=json.loads(open('$O/bench_$v.json').read()); print('$v', round(d['ms_per_step'],4), round(d['value']), d.get('posemb_status',{}).get('flags'), round(d['stage_rooflines']['gin_encoder_fwd']['ms_in_step'],3), round(d['stage_rooflines']['gin_encoder_bwd']['ms_in_step'],3))"
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
