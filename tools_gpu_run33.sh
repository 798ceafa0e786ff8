#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/gin_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ginphases33.log | tail -5
exit 0
