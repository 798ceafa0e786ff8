#!/bin/bash
# Variant builds under the launch trace (training stream alone): 16 gathered rows in flight at 2 workgroups per CU, slabs per round in nce_dq.
set -u
O=gpurun_out/r3c40
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for v in default j16 dq4 dq8; do
  lib=""; [ $v != default ] && lib="--lib $GRAFT_REPO_ROOT/gcc_amd/csrc/variants/lib_$v.so"
  rm -rf /tmp/tr_$v
  cd /tmp && (timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 $lib 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/run_$v.log; cd $GRAFT_REPO_ROOT
  (python tools/stream_trace.py /tmp/tr_$v 2>&1) > $O/stream_trace_$v.txt
  echo "== $v: $(cat $O/run_$v.log)"
  grep -E "^busy|gin_in_kernel  |gin_bwd_c_kernel  |gin_bwd_emb_kernel  |nce_dq_kernel  " $O/stream_trace_$v.txt
done
