#!/bin/bash
# Round 3, call 12: heavy-phase gate across producer lanes (gcc_posemb_multi_gated): pipeline parity test, bench at the
# driver's flags and sustained, gate on / off, gated caps.
set -u
O=gpurun_out/r3c12
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_posemb_gpu.py tests/test_rccl_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest.log
tail -3 $O/pytest.log
run() { (timeout 300 python bench.py --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['stage_ms'].get('gin_fwd'),3), round(d['stage_ms'].get('gin_bwd'),3), [round(v,1) for k,v in d['stage_ms'].items() if k.startswith('posemb')])") 2>&1 | tail -1; }
for steps in 20 160; do
  echo "[gate off steps $steps] $(GCC_POSEMB_GATE=0 run --steps $steps)" | tee -a $O/sweep.txt
  for gc in "128,128" "96,128" "128,192" "192,192" "64,96"; do
    echo "[gate on caps $gc steps $steps] $(GCC_POSEMB_GATED_CAPS=$gc run --steps $steps)" | tee -a $O/sweep.txt
  done
done
