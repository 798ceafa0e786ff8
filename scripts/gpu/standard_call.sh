#!/bin/bash
# The measurements every change of round 3 was judged by, as one parametrised call (each part takes 5-20 s on the box):
#   bash scripts/gpu/standard_call.sh <out-tag> [tests] [stream] [phases] [bench] [driver] [variant:<name>] ...
#     tests    pytest -m gpu of the encoder / step / pipeline / posemb files          -> pytest.log
#     stream   tools/graph_probe.py + rocprofv3 launch trace + tools/stream_trace.py  -> probe.txt, stream_trace.txt
#     phases   tools/posemb_phases.py (16-view eigensolver call, CU-time per class)    -> phases.txt
#     bench    bench.py --steps 192 --warmup 64 --no-cpu-baseline (sustained)          -> bench_192.json
#     driver   bench.py --steps 20 --warmup 5 --no-cpu-baseline (the driver's flags)   -> bench_driver.json
#     variant:<name>  swap gcc_amd/csrc/variants/lib_<name>.so (tools/build_variant.sh) in for the parts that FOLLOW it;
#                     variant:default swaps the product build back (also done at the end)
set -u
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
v=default
for part in "$@"; do
  case $part in
    variant:*) v=${part#variant:}
      if [ $v = default ]; then cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so; else cp gcc_amd/csrc/variants/lib_$v.so gcc_amd/csrc/libgcc_amd.so; fi ;;
    tests) (timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_train_step_gpu.py tests/test_pipeline_gpu.py tests/test_posemb_gpu.py -m gpu -q --tb=short 2>&1 | tail -30) > $O/pytest_$v.log
      echo "[$v] tests: $(grep -E 'passed|failed' $O/pytest_$v.log)" ;;
    stream) (timeout 300 python tools/graph_probe.py --steps 200 2>&1 | tail -2) > $O/probe_$v.txt
      rm -rf /tmp/tr_$v
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/tools/graph_probe.py --steps 60 > /dev/null 2>&1)
      (python tools/stream_trace.py /tmp/tr_$v 2>&1) > $O/stream_trace_$v.txt
      echo "[$v] $(tail -n 1 $O/probe_$v.txt)"; grep -E "^busy" $O/stream_trace_$v.txt; sed -n '/^busy/,$p' $O/stream_trace_$v.txt | tail -n +2 | head -24 ;;
    phases) (timeout 300 python tools/posemb_phases.py 2>&1 | grep -E "multi call|^mid|^cheb|^wave|^total|status") > $O/phases_$v.txt
      echo "[$v]"; cut -c1-250 $O/phases_$v.txt ;;
    bench|driver)
      if [ $part = bench ]; then flags="--steps 192 --warmup 64"; f=bench_192; else flags="--steps 20 --warmup 5"; f=bench_driver; fi
      (timeout 400 python bench.py $flags --no-cpu-baseline 2>>$O/bench_$v.err | tail -1) > $O/${f}_$v.json
      python -c "
import json; d=json.loads(open('$O/${f}_$v.json').read()); s=d['stage_rooflines']; print('[$v] $f', round(d['ms_per_step'],4), 'ms/step', round(d['value']), 'subgraphs/s flags', (d.get('posemb_status') or {}).get('flags'), 'encoder fwd/bwd in step', round(s['gin_encoder_fwd']['ms_in_step'],3), round(s['gin_encoder_bwd']['ms_in_step'],3))" ;;
    *) echo "unknown part $part" ;;
  esac
done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
