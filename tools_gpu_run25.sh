#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for q in 4 3; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q PROBE_KINDS=kry6,slot14,mid,small PROBE_STREAMS=2,3 timeout 400 python tools/contention_probe.py 2>&1 | grep "bg streams"
done 2>&1 | tee gpurun_out/contention25.log
exit 0
