#!/bin/bash
# Wide GIN, final of the round: tests, roofline line (both kernel shapes), ablations of the new shape, SQ counters.
set -u
O=gpurun_out/r3c32
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gin_wide_gpu.py -m gpu -q --tb=short 2>&1 | tail -15) > $O/pytest.log
grep -E "passed|failed" $O/pytest.log
(GCC_GINW_KERNEL=1 timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/gin_roofline_c5_shape1.json
(timeout 600 python tools/gin_roofline.py --phases 2>/dev/null | tail -1) > $O/gin_roofline_c5.json
for d in 1 2 4 7; do
(GCC_GINW_DBG=$d timeout 600 python tools/gin_roofline.py 2>/dev/null | tail -1) > $O/ablate_dbg$d.json
done
python - <<PY
import json
for f in ["gin_roofline_c5_shape1","gin_roofline_c5","ablate_dbg1","ablate_dbg2","ablate_dbg4","ablate_dbg7"]:
    d=json.loads(open("$O/"+f+".json").read())
    print(f, "fused ms %.3f TFLOP/s %.0f frac %.3f layerwise ms/layer %.3f" % (d['fused']['ms'], d['fused']['tflops'], d['fused']['roofline']['frac'], d['layerwise']['ms_per_layer']), {a: round(b,1) for a,b in d.get('fused_phase_us_per_subgraph',{}).items()})
PY
cd /tmp && (timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pmc_g -o g -- python $GRAFT_REPO_ROOT/tools/gin_roofline.py --iters 2 --warmup 1 2>&1 | tail -2) > $GRAFT_REPO_ROOT/$O/pmc_g.log; cd $GRAFT_REPO_ROOT
(timeout 60 python tools/pmc_gin_wide.py /tmp/pmc_g $O/pmc_gin_wide.json 2>&1 | tail -30) > $O/pmc_gin_wide.log
tail -12 $O/pmc_gin_wide.log
cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/st_g -o s -- python $GRAFT_REPO_ROOT/tools/gin_roofline.py --iters 20 --warmup 5 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/stats_run.log; cd $GRAFT_REPO_ROOT
find /tmp/st_g -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_gin_wide.csv
head -4 $O/kernel_stats_gin_wide.csv | cut -c1-200
