#!/bin/bash
# Wide GIN: third kernel shape (two passes per product, the first pass's epilogue between the second pass's matrix
# instructions) against the second.
set -u
O=gpurun_out/r3c36
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gin_wide_gpu.py -m gpu -q --tb=short 2>&1 | tail -15) > $O/pytest.log
grep -E "passed|failed" $O/pytest.log
(GCC_GINW_KERNEL=2 timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/gin_roofline_c5_shape2.json
(timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/gin_roofline_c5.json
(timeout 600 python tools/gin_roofline.py 2>/dev/null | tail -1) > $O/gin_roofline_c5_b.json
python - <<PY
import json
for f in ["gin_roofline_c5_shape2","gin_roofline_c5","gin_roofline_c5_b"]:
    d=json.loads(open("$O/"+f+".json").read())
    print(f, "fused ms %.3f TFLOP/s %.0f frac %.3f layerwise ms/layer %.3f" % (d['fused']['ms'], d['fused']['tflops'], d['fused']['roofline']['frac'], d['layerwise']['ms_per_layer']), {a: round(b,1) for a,b in d.get('fused_phase_us_per_subgraph',{}).items()})
PY
