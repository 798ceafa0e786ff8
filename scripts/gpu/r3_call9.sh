#!/bin/bash
# Round 3, call 8 (re-run as call 9 with the one-wave register Cholesky of the sparse block class)
# launcher path once (oversubscribed on one GPU: gloo staging, correctness only).
set -u
O=gpurun_out/r3c9
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_posemb_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest_posemb.log
tail -3 $O/pytest_posemb.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases.txt
cat $O/posemb_phases.txt
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench.json; cat $O/bench.json
