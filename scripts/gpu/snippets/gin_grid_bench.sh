# bench (in the pipeline, next to the producers) by the tile kernels' grid (GCC_GIN_GRID: workgroups per pass; default = the rows' tiles, <= 768)
for g in ${GRIDS:-0 384 448 512 576 640}; do
  for flags in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 192 --warmup 64"; do
    if [ $g = 0 ]; then unset GCC_GIN_GRID; else export GCC_GIN_GRID=$g; fi
    r=$(timeout 400 python bench.py $flags --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_rooflines']; print(round(d['ms_per_step'],4), [round(x,3) for x in d.get('ms_per_step_windows',[])], 'fwd/bwd in step', round(s['gin_encoder_fwd']['ms_in_step'],3), round(s['gin_encoder_bwd']['ms_in_step'],3))")
    echo "GCC_GIN_GRID=$g | $flags | $r"
  done
done | tee $O/gin_grid_bench.txt
