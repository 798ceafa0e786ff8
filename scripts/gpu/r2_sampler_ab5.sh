#!/bin/bash
# induction: 512- and 1024-thread workgroups (16 / 32 units per virtual workgroup), with and without 8 waves per SIMD forced
set -u
O=gpurun_out/r2ab5
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() {
  cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
}
build() {
  (cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
}
G2="--nodes 10000000 --edges 200000000 --launches 40"
build -DGCC_INDUCE_THREADS=512
(timeout 200 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -1) > $O/pytest_t512.txt
run t512_g1 "--launches 60"
run t512_g2 "$G2"
build -DGCC_INDUCE_THREADS=512 "-DGCC_INDUCE_OCC=__attribute__((amdgpu_waves_per_eu(8,8)))"
run t512o8_g1 "--launches 60"
run t512o8_g2 "$G2"
build -DGCC_INDUCE_THREADS=1024 "-DGCC_INDUCE_OCC=__attribute__((amdgpu_waves_per_eu(8,8)))"
(timeout 200 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -1) > $O/pytest_t1024.txt
run t1024o8_g1 "--launches 60"
run t1024o8_g2 "$G2"
build -DGCC_INDUCE_THREADS=256 "-DGCC_INDUCE_OCC=__attribute__((amdgpu_waves_per_eu(8,8)))"
run t256o8_g2 "$G2"
cat $O/pytest_*.txt
