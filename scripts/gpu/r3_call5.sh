#!/bin/bash
# Round 3, call 5: wave teams with the register-resident tridiagonalisation and the prefetching back-transformation;
# w48 built for 2 waves per SIMD (a few spills) vs 1 (no spills).
set -u
O=gpurun_out/r3c5
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_posemb_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest_posemb.log
tail -3 $O/pytest_posemb.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -4) > $O/posemb_phases_occ2.txt
cat $O/posemb_phases_occ2.txt
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench_occ2.json; cat $O/bench_occ2.json
(cd gcc_amd/csrc && touch posemb.hip && make EXTRA=-DGCC_POSEMB_W48_OCC=1 2>&1 | grep -E "\berror\b")
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -4) > $O/posemb_phases_occ1.txt
cat $O/posemb_phases_occ1.txt
(timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench_occ1.json; cat $O/bench_occ1.json
