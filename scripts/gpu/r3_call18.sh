#!/bin/bash
# Round 3, call 18: start records of the induce workgroups by their own multi-workgroup kernel (was the tail of the
# single-workgroup prefix kernel): sampler GPU tier, sampler mode on both graphs, kernel stats of a 16-step launch.
set -u
O=gpurun_out/r3c18
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
(timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_pipeline_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest.log
grep -E "passed|failed|Error" $O/pytest.log | head -5
run() { (timeout 600 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), d.get('stage_rooflines',{}).get('sampler_end_to_end',{}).get('frac'), d['kernel_ms_isolated'])") 2>&1 | tail -1; }
echo "[sampler G1] $(run --mode sampler --nodes 1000000 --edges 10000000 --steps 96 --warmup 16)" | tee -a $O/sweep.txt
echo "[sampler G2] $(run --mode sampler --steps 96 --warmup 16)" | tee -a $O/sweep.txt
echo "[train 20] $(run --steps 20 --warmup 5)" | tee -a $O/sweep.txt
