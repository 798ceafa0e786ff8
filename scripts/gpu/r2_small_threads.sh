#!/bin/bash
# eigensolver small class (n <= 64, 73 % of the items): threads per item (256 default; 128, 64, 512, 1024 tried), with and without more resident workgroups
set -u
O=gpurun_out/r2st2
mkdir -p $O
export TMPDIR=/tmp
build() {
  (cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
}
for t in 512 1024; do
  build -DGCC_POSEMB_SMALL_T=$t
  (timeout 200 python tools/posemb_phases.py 2>&1 | tail -9 | head -3) > $O/phases_t$t.txt
  (timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/err.txt | tail -1) > $O/bench_t${t}_caps256.json
  (GCC_POSEMB_GRID_CAPS=512,128,128,64,64,128 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/err.txt | tail -1) > $O/bench_t${t}_caps512.json
done
(timeout 300 python -m pytest tests/test_posemb_gpu.py -q -m gpu 2>&1 | tail -2) > $O/pytest_t1024.txt
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); print(d['ms_per_step'], d['value'], d['posemb_status']['flags'], d['stage_ms'])"; done
cat $O/phases_t*.txt $O/pytest_t1024.txt
