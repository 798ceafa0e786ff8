#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for q in 2 4 8 32; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q PROBE_KINDS=slot2 PROBE_STREAMS=2,8 timeout 300 python tools/contention_probe.py 2>&1 | grep "bg streams"
done 2>&1 | tee gpurun_out/contention24.log
exit 0
