#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools_hunt.py > gpurun_out/hunt.log 2>&1
tail -4 gpurun_out/hunt.log
