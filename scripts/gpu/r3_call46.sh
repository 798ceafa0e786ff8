#!/bin/bash
# How long does the host take to issue one training step?
set -u
O=gpurun_out/r3c46
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -2) > $O/probe.txt
cat $O/probe.txt
