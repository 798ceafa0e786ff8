#!/bin/bash
# Round 3, call 17: gcc_sample_multi (a producer chunk's batches in one launch set): GPU parity (multi-step pipelined run
# vs single-step sequential run, sampler tier), bench at the driver's flags + sustained, sampler mode on both graphs by
# steps per call.
set -u
O=gpurun_out/r3c17
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py tests/test_rccl_gpu.py tests/test_train_main_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest.log
grep -E "passed|failed|Error" $O/pytest.log | head -5
run() { (timeout 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('stage_ms',{}); print(round(d['ms_per_step'],4), round(d['value']), 'sampler', round(s.get('sampler',0),3), 'fwd', round(s.get('gin_fwd',0),3), 'bwd', round(s.get('gin_bwd',0),3), d.get('stage_rooflines',{}).get('sampler_end_to_end',{}).get('frac'))") 2>&1 | tail -1; }
echo "[train steps 20] $(run --steps 20 --warmup 5)" | tee -a $O/sweep.txt
echo "[train steps 160] $(run --steps 160 --warmup 5)" | tee -a $O/sweep.txt
echo "[train steps 192] $(run --steps 192 --warmup 64)" | tee -a $O/sweep.txt
for S in 1 4 8 16; do
  echo "[sampler G1 steps/call $S] $(run --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16 --sampler-steps $S)" | tee -a $O/sweep.txt
done
for S in 1 8 16; do
  echo "[sampler G2 steps/call $S] $(run --mode sampler --steps 48 --warmup 16 --sampler-steps $S)" | tee -a $O/sweep.txt
done
