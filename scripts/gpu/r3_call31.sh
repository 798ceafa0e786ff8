#!/bin/bash
# Wide GIN second kernel shape: where does a product's time go?  Ablations (wrong results, timing only): GCC_GINW_DBG
# 1 no epilogue arithmetic, 2 no weight requests in the products, 4 no fragment reads in the k loops, 7 all three.
set -u
O=gpurun_out/r3c31
mkdir -p $O
export TMPDIR=/tmp
for d in 16 0; do
(GCC_GINW_DBG=$d timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/roofline_dbg$d.json
python - <<PY
import json
d=json.loads(open("$O/roofline_dbg$d.json").read())
print("dbg $d: fused ms %.3f TFLOP/s %.0f" % (d['fused']['ms'], d['fused']['tflops']), {a: round(b,1) for a,b in d['fused_phase_us_per_subgraph'].items()})
PY
done
