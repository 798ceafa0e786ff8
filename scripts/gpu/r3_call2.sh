#!/bin/bash
# Round 3, call 2: one-wave eigensolver teams (n' <= 48 / <= 64) -- GPU parity tier of the solver, phase ticks with the
# wave classes on and off, bench at the driver's flags.
set -u
O=gpurun_out/r3c2
mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_posemb_gpu.py tests/test_pipeline_gpu.py tests/test_train_main_gpu.py -q --tb=short -x 2>&1 | tail -30) > $O/pytest_posemb.log
tail -3 $O/pytest_posemb.log
(timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_wave.txt
cat $O/posemb_phases_wave.txt
(GCC_POSEMB_WAVE=0 timeout 200 python tools/posemb_phases.py 2>&1 | tail -12) > $O/posemb_phases_nowave.txt
head -3 $O/posemb_phases_nowave.txt
(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_driver.err | tail -1) > $O/bench_driver.json
cut -c1-260 $O/bench_driver.json
for caps in "256,128,128,64,64,128,64,32" "256,128,128,64,64,128,256,128"; do
  (GCC_POSEMB_GRID_CAPS=$caps timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200) > $O/bench_caps_${caps//,/_}.json
  echo "caps $caps"; cat $O/bench_caps_${caps//,/_}.json
done
tail -n 5 $O/bench_driver.err
