#!/bin/bash
# what gin_in_kernel's 79 us are made of: the kernel with one part removed at a time (timing only; results are wrong)
set -u
O=gpurun_out/r2abl
mkdir -p $O
export TMPDIR=/tmp
for v in 1 2 4 7; do
  (cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGIN_DBG_SKIP=$v -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
  cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$v -o i -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --lanes 1 --chunk 1 --posemb placeholder >/dev/null 2>&1); cd $GRAFT_REPO_ROOT
  find /tmp/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_skip$v.csv
  echo "skip $v: $(grep gin_in_kernel $O/kernel_stats_skip$v.csv | cut -d, -f2-5)"
done
