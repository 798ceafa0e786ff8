#!/bin/bash
# Round 3, call 3: one-wave teams after the ILP pass (batched LDS loads in the tridiagonalisation, two Sturm chains per
# lane, register-resident back-transformation, two nodes per expansion step): solver tests, phase ticks, bench by grid caps.
set -u
O=gpurun_out/r3c3
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_posemb_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest_posemb.log
tail -3 $O/pytest_posemb.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_wave.txt
cat $O/posemb_phases_wave.txt
for caps in "256,128,128,64,64,128,128,64" "256,128,128,64,64,128,256,128" "256,128,128,64,64,128,512,128"; do
  (GCC_POSEMB_GRID_CAPS=$caps timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench_caps_${caps//,/_}.json
  echo "caps $caps"; cat $O/bench_caps_${caps//,/_}.json
done
