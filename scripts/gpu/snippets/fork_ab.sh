# bench at the driver's flags and sustained with the eigensolver classes forked onto side streams (GCC_POSEMB_FORK: 0 one stream, 1 three-way, 2 the block class beside the rest)
for f in ${FORKS:-0 1 2}; do
  for flags in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 192 --warmup 64"; do
    r=$(GCC_POSEMB_FORK=$f timeout 400 python bench.py $flags --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), [round(x,3) for x in d.get('ms_per_step_windows',[])])")
    echo "fork $f | $flags | $r"
  done
done | tee $O/fork_ab.txt
