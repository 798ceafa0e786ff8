#!/bin/bash
# The 65..128 class of the eigensolver with 512 / 256 threads per workgroup instead of 1024 (its barriers are 16-wave barriers): phases of a 16-view call.
set -u
O=gpurun_out/r3c44
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
for v in default mid512 mid256; do
  if [ $v = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; fi
  (timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^total|status") > $O/phases_$v.txt
  echo "== $v"; cut -c1-230 $O/phases_$v.txt
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
