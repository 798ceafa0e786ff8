#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/posemb_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phases28.log | tail -12
exit 0
