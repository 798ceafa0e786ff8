# the driver's 20-step window by eigensolver grid caps (three processes each)
run() { echo "$1: $(for i in 1 2 3; do env GCC_POSEMB_GRID_CAPS=$1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), [round(x,3) for x in d["ms_per_step_windows"][1:]], end="  ")'; done)"; }
( run 256,64,128,64,64,96,512,128,128
  run 256,64,128,64,64,64,128,64,128
  run 256,64,128,64,64,96,128,64,128
  run 256,64,128,64,64,96,64,32,128
  run 256,64,128,64,64,64,128,64,96 ) | tee $O/caps_driver.txt
