# grid barrier vs kernel boundary (tools/probes/grid_barrier_probe.hip; the binary is built locally and travels)
(timeout 120 tools/probes/grid_barrier_probe 2>&1) > $O/grid_barrier_probe.txt; cat $O/grid_barrier_probe.txt
