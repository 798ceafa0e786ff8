#!/bin/bash
# induction: units per virtual workgroup 8 vs 4 (one unit in flight per wave), on G1 and on the 10M/200M graph
set -u
O=gpurun_out/r2ab2
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
run() {
  cd /tmp && (timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py $2 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
}
build() {
  (cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $1 -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
}
G2="--nodes 10000000 --edges 200000000 --launches 40"
run v8_g1 "--launches 60"
run v8_g2 "$G2"
build "-DGCC_INDUCE_VWG_UNITS=4"
run v4_g1 "--launches 60"
run v4_g2 "$G2"
(timeout 120 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -2) > $O/pytest_v4.txt
build "-DGCC_INDUCE_VWG_UNITS=16"
run v16_g2 "$G2"
cat $O/pytest_v4.txt
