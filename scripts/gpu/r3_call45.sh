#!/bin/bash
# Does the sampler's occupancy matter to the step?  Builds whose sampler launches reserve 32 / 64 KiB of unused dynamic LDS per workgroup
# (fewer resident sampler workgroups per CU; the sampler lane has ~20x headroom), swapped in on the box only: sampler lanes alone and the full pipeline.
set -u
O=gpurun_out/r3c45
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
for v in default pad32768 pad65536; do
  if [ $v = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; fi
  for m in placeholder full; do
    extra=""; [ $m = placeholder ] && extra="--posemb placeholder"
    (timeout 300 python bench.py --steps 192 --warmup 64 --no-cpu-baseline $extra 2>>$O/bench.err | tail -1) > $O/bench_${v}_$m.json
    python -c "
import json; d=json.loads(open('$O/bench_${v}_$m.json').read()); print('$v', '$m', round(d['ms_per_step'],4), round(d['value']), (d.get('posemb_status') or {}).get('flags'), round(d['stage_rooflines']['gin_encoder_fwd']['ms_in_step'],3), round(d['stage_rooflines']['gin_encoder_bwd']['ms_in_step'],3), d.get('stage_ms',{}).get('sampler'))"
  done
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
