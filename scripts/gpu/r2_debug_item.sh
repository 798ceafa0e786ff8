#!/bin/bash
# debug build on the GPU box's scratch copy only
set -u
O=gpurun_out/r2dbg
mkdir -p $O
cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGCC_POSEMB_DEVDEBUG -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep -E " error" ; cd ../..
(timeout 120 python tests/tools/posemb_item_debug.py tests/golden/posemb_item_s4_v1_b126.npz 3532987934 2>&1 | tail -2000) > $O/item_trace.txt
tail -5 $O/item_trace.txt
