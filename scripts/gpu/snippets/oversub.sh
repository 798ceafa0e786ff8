# the N > 1 launch path of bench.py on ONE GPU (ranks share the device, gloo staging: a correctness run, not a number)
for n in 2 8; do
  (timeout 900 python bench.py --gpus $n --steps 8 --warmup 2 --no-cpu-baseline --no-parity 2>$O/bench_gpus$n.err | tail -1) > $O/bench_gpus${n}_oversubscribed_on_1gpu.json
  python -c "
import json
try:
    d=json.load(open('$O/bench_gpus${n}_oversubscribed_on_1gpu.json')); print('gpus $n:', d['n_gpus'], 'ranks', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'subgraphs/s;', d['config']['parallelism'][:90])
except Exception as e:
    print('gpus $n FAILED', e); import subprocess; print(subprocess.run('grep -v \"^frame\" $O/bench_gpus$n.err | grep -i \"error\|Traceback\" | head -5 | cut -c1-250', shell=True, capture_output=True, text=True).stdout)
"
done | tee $O/oversub.txt
