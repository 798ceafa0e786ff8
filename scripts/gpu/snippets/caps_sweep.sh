# bench at the driver's flags and sustained, by the eigensolver classes' grid caps (GCC_POSEMB_GRID_CAPS = small,mid,slot,krylov,big,cheb,w48,w64,pair)
for caps in ${CAPS:-"256,64,128,64,64,96,128,64,128" "256,64,128,64,64,64,128,64,128" "256,64,128,64,64,96,64,32,128" "256,64,128,64,64,64,64,32,128" "256,64,128,64,64,128,128,64,128" "256,64,128,64,64,96,128,64,64"}; do
  for flags in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 192 --warmup 64"; do
    r=$(GCC_POSEMB_GRID_CAPS=$caps timeout 400 python bench.py $flags --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,3) for x in d.get('ms_per_step_windows',[])])")
    echo "caps $caps | $flags | $r"
  done
done | tee $O/caps_sweep.txt
