#!/bin/bash
# induction with the per-unit row strip, 24-bit hash: 256 / 512 / 768-thread workgroups on G1 and on the 10M/200M graph
set -u
O=gpurun_out/r2ab8
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
for t in 256 512; do
  build -DGCC_INDUCE_THREADS=$t
  (timeout 200 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -1) > $O/pytest_t$t.txt
  run t${t}_g1 "--launches 60"
  run t${t}_g2 "$G2"
done
(timeout 300 python tools/induce_phases.py 2>&1 | tail -1) > $O/phases_t512_g1.txt
cat $O/pytest_*.txt $O/phases*.txt
