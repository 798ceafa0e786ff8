#!/bin/bash
# Round 3, call 1: GPU tier with the new parity tests (Philox dropout, pipelined vs sequential producer, fused E2E step,
# per-shard sampling), bench at the driver's flags (new stage_rooflines / reference_shaped), --mode e2e at bsz 256 and 32,
# eigensolver phase ticks as the baseline for this round's solver work.
set -u
O=gpurun_out/r3c1
mkdir -p $O
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -40) > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
(timeout 500 python bench.py --steps 20 --warmup 5 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
cut -c1-300 $O/bench_driver.json
(timeout 300 python bench.py --mode e2e --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_e2e256.err | tail -1) > $O/bench_e2e256.json
cut -c1-200 $O/bench_e2e256.json
(timeout 300 python bench.py --mode e2e --batch-size 32 --steps 40 --warmup 10 --cpu-seconds 8 2>$O/bench_e2e32.err | tail -1) > $O/bench_e2e32.json
cut -c1-200 $O/bench_e2e32.json
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -9) > $O/posemb_phases.txt
cat $O/posemb_phases.txt
tail -5 $O/bench_driver.err $O/bench_e2e256.err $O/bench_e2e32.err
