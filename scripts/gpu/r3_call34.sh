#!/bin/bash
# gather_tile ablations (library builds with -DGATHER_DBG=1 no side-slot pass, 2 no row search, 4 no feature loads; wrong
# results, timing only): the training stream alone and gin_in / gin_bwd_c / gin_bwd_emb per launch.
set -u
O=gpurun_out/r3c34
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for v in 0 1 2 4; do
L=""; [ $v != 0 ] && L="--lib $GRAFT_REPO_ROOT/gpurun_variants_libg$v.so"
cd /tmp && rm -rf /tmp/trg && (timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 $L 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/trace$v.log; cd $GRAFT_REPO_ROOT
echo "GATHER_DBG=$v: $(tail -1 $O/trace$v.log)"
(python tools/stream_trace.py /tmp/trg 2>&1 | grep -E "^  gin_in|^  gin_bwd_c|^  gin_bwd_emb")
done
