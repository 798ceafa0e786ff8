#!/bin/bash
# Wide GIN (config 5): second kernel shape (one wave per SIMD, 64 x 128 register tiles) against the first.
set -u
O=gpurun_out/r3c30
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gin_wide_gpu.py -m gpu -q --tb=short 2>&1 | tail -15) > $O/pytest.log
grep -E "passed|failed|Error|error" $O/pytest.log | head
(GCC_GINW_KERNEL=1 timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/roofline_k1.json
(timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/roofline_k2.json
for k in k1 k2; do python -c "
import json; d=json.loads(open('$O/roofline_$k.json').read()); print('$k', {x: d[x] for x in d if x in ('fused_ms','tflops','frac_of_peak','per_layer_ms','phases_us_per_subgraph','fused_tflops','frac')} or d)"; done
