# sustained bench (192 steps) by producer lanes and eigensolver grid caps (small,mid,slot,krylov,big,cheb,w48,w64,pair)
run() { echo "$1 | $2: $(env $2 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --no-parity $1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), (d.get("posemb_status") or {}).get("flags"))')"; }
( run "" "X=1"
  run "--lanes 3" "X=1"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,128,512,128,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,160,512,128,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,512,128,192"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128,128"
  run "--lanes 3" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,512,128,96"
  run "--chunk 8" "X=1"
  run "" "X=1" ) | tee $O/caps_sweep.txt
