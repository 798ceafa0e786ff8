# the multi-GPU launch path on one rank, repeatedly: N runs of 1024 steps with the RCCL hand-offs between the step's graph segments
for i in 1 2 3 4 5 6; do
  (env ${SOAK_ENV:-X=1} timeout 600 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline --no-parity --collectives 2>$O/soak_$i.err | tail -1) > $O/soak_$i.json
  python -c "
import json
try:
    d=json.load(open('$O/soak_$i.json')); print('soak $i:', round(d['ms_per_step'],4), 'ms/step; graph_capture_failures', d.get('graph_capture_failures'), 'replays', d.get('graph_replays_in_timed_region'))
except Exception as e:
    print('soak $i FAILED', e); import subprocess; print(subprocess.run('grep -v \"^frame\" $O/soak_$i.err | grep -i \"error\|warn\" | head -5 | cut -c1-250', shell=True, capture_output=True, text=True).stdout)
"
done | tee $O/coll_soak.txt
