# sustained bench by the one-wave classes' grid caps (w48, w64) and the small / slot classes (positions: small,mid,slot,krylov,big,cheb,w48,w64,pair)
run() { echo "$1 | $2: $(env $2 timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --no-parity $1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); s=d["stage_ms"]; print(round(d["ms_per_step"],4), "fwd/bwd in step", round(s["gin_fwd"],3), round(s["gin_bwd"],3), "posemb chunk", round(s.get("posemb_chunk_of_32_views",0),2))')"; }
( run "" "X=1"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,256,128,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,256,64,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,128,64,128"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,64,256,64,96"
  run "" "GCC_POSEMB_GRID_CAPS=128,64,64,64,64,64,256,64,96"
  run "" "GCC_POSEMB_GRID_CAPS=256,64,128,64,64,96,768,192,128"
  run "" "X=1" ) | tee $O/caps_sweep2.txt
