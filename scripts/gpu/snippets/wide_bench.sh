# the any-width step (train.py:93 --hidden-size above 64): device test tier of the wide path, train.py end to end, bench lines at 128 / 256
(timeout 900 python bench.py --hidden-size 256 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_hidden256.err | tail -1) > $O/bench_hidden256.json
python -c "
import json; d=json.loads(open('$O/bench_hidden256.json').read()); r=d['roofline']; print('hidden 256:', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'subgraphs/s | roofline', r['kernel'][:30], 'achieved', round(r['achieved'],2), 'TF frac', round(r['frac'],4), 'stream ms', round(r['training_stream_ms_isolated'],3), '| parity', list((d.get('parity') or {}).keys()))" || tail -5 $O/bench_hidden256.err
(timeout 900 python bench.py --hidden-size 128 --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>$O/bench_hidden128.err | tail -1) > $O/bench_hidden128.json
python -c "
import json; d=json.loads(open('$O/bench_hidden128.json').read()); r=d['roofline']; print('hidden 128:', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'subgraphs/s | achieved', round(r['achieved'],2), 'TF frac', round(r['frac'],4), 'stream ms', round(r['training_stream_ms_isolated'],3))" || tail -5 $O/bench_hidden128.err
