#!/bin/bash
# feature rows in flight per lane group of the edge-balanced gather: 4 / 6 (8 is the default; it spills 23 VGPRs in gin_in_kernel)
set -u
O=gpurun_out/r2gj
mkdir -p $O
export TMPDIR=/tmp
for j in 4 6; do
  (cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGCC_GATHER_J=$j -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
  cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$j -o i -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --lanes 1 --chunk 1 --posemb placeholder >/dev/null 2>&1); cd $GRAFT_REPO_ROOT
  find /tmp/prof_$j -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_j$j.csv
  echo "J=$j: $(grep -E 'gin_in_kernel|gin_bwd_c_kernel|gin_bwd_emb' $O/kernel_stats_j$j.csv | cut -d, -f2-4 | tr '\n' ' ')"
done
