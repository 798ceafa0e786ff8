#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=16
timeout 600 python tools/contention_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/contention23.log | tail -20
exit 0
