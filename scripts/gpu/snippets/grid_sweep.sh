# training stream alone (tools/graph_probe.py) by the tile kernels' grid (GCC_GIN_GRID: workgroups per pass; default = capacity, <= 768)
for g in ${GRIDS:-416 448 480 512 640 768}; do
  echo "GCC_GIN_GRID=$g $(GCC_GIN_GRID=$g timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -1)"
done | tee $O/grid_sweep.txt
