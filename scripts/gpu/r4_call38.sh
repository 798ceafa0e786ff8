#!/bin/bash
# Round 4, call 38: tools/load_probe.py -- the training stream next to synthetic co-tenants (occupancy only / memory traffic / arithmetic)
set -u
O=gpurun_out/r4c38
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 600 python tools/load_probe.py 2>&1 | tail -24) | tee $O/load_probe.txt
