#!/bin/bash
# Training stream alone (tools/graph_probe.py): launch-by-launch trace of one step; A/B of the BatchNorm totals path
# (GCC_BN_TOTALS=0: consumers add the 32 replicas up, the producing kernels take no arrival ticket).
set -u
O=gpurun_out/r3c20
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) > $O/probe_totals1.txt
(GCC_BN_TOTALS=0 timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1) > $O/probe_totals0.txt
cat $O/probe_totals1.txt $O/probe_totals0.txt
cd /tmp && (timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/trace_run.log; cd $GRAFT_REPO_ROOT
(python tools/stream_trace.py /tmp/tr 2>&1) > $O/stream_trace.txt
head -5 $O/stream_trace.txt; tail -45 $O/stream_trace.txt
