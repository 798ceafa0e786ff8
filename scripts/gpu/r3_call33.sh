#!/bin/bash
# gin_in_kernel ablations (library builds with -DGIN_DBG_SKIP=1 no pooling, 2 no statistics flush, 4 no gather; wrong
# results, timing only): the training stream alone with each.
set -u
O=gpurun_out/r3c33
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) | tee $O/full.txt
for v in 1 2 4; do
echo "GIN_DBG_SKIP=$v"; (timeout 300 python tools/graph_probe.py --steps 200 --lib gpurun_variants_libskip$v.so 2>&1 | tail -1) | tee $O/skip$v.txt
done
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 --lib $GRAFT_REPO_ROOT/gpurun_variants_libskip1.so 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/trace1.log; cd $GRAFT_REPO_ROOT
(python tools/stream_trace.py /tmp/tr1 2>&1 | grep -E "gin_in|median") | head -8
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr4 -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 --lib $GRAFT_REPO_ROOT/gpurun_variants_libskip4.so 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/trace4.log; cd $GRAFT_REPO_ROOT
(python tools/stream_trace.py /tmp/tr4 2>&1 | grep -E "gin_in|median") | head -8
