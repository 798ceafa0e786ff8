#!/bin/bash
# gin_in_kernel at the current build: phase ticks, and timing-only builds without pooling (1) / statistics flush (2) / gather (4) / all three (7).
set -u
O=gpurun_out/r3c41
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 300 python tools/gin_phases.py 2>&1 | tail -2) > $O/gin_phases.txt; cat $O/gin_phases.txt
for v in skip1 skip2 skip4 skip7; do
  lib="--lib $GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_$v.so"
  rm -rf /tmp/tr_$v
  cd /tmp && (timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 $lib 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/run_$v.log; cd $GRAFT_REPO_ROOT
  (python tools/stream_trace.py /tmp/tr_$v 2>&1) > $O/stream_trace_$v.txt
  echo "== $v"
  grep -E "gin_in_kernel  " $O/stream_trace_$v.txt
  grep -E "gin_in_kernel$" $O/stream_trace_$v.txt | head -4
done
