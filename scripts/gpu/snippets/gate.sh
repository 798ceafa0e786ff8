# the heavy-phase gate across producer lanes (GCC_POSEMB_GATE=1: the lanes' eigensolver calls take turns), window and sustained
for e in "GCC_POSEMB_GATE=0" "GCC_POSEMB_GATE=1"; do
  echo "$e window: $(for i in 1 2 3; do env $e timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), end=" ")'; done)"
  echo "$e sustained: $(for i in 1 2; do env $e timeout 400 python bench.py --steps 192 --warmup 64 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), end=" ")'; done)"
done | tee $O/gate.txt
