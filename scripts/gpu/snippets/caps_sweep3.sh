# sustained bench around (cheb, w48, w64, pair) = (96|64, 128, 64, 128|96), with 2 and 3 lanes
run() { echo "$1 | $2: $(env $2 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --no-parity $1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print(round(d["ms_per_step"],4), "fwd/bwd in step", round(s["gin_fwd"],3), round(s["gin_bwd"],3), "posemb chunk", round(s.get("posemb_chunk_of_32_views",0),2))')"; }
( run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,64,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,96,48,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,64,32,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,128,64,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,128,64,96"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,64,96"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,80,128,64,112"
  run "--lanes 3" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,96,48,96"
  run "--lanes 3" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,64,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,64,128" ) | tee $O/caps_sweep3.txt
