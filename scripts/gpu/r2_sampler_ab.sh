#!/bin/bash
# induction: both units of a wave in flight (111 VGPRs, 4 workgroups per CU) vs one at a time (more resident workgroups)
set -u
O=gpurun_out/r2ab
mkdir -p $O
export TMPDIR=/tmp
run() {
  cd /tmp && (timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_$1 -o s -- python $GRAFT_REPO_ROOT/tools/sampler_alone.py --launches 60 2>&1 | tail -1) > $GRAFT_REPO_ROOT/$O/log_$1.txt; cd $GRAFT_REPO_ROOT
  find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$1.csv
}
run inflight2
(cd gcc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGCC_INDUCE_INFLIGHT=1 -o libgcc_amd.so common.hip sampler.hip encoder.hip encoder_bwd.hip nce.hip posemb.hip gin_wide.hip 2>&1 | grep " error")
run inflight1
(timeout 120 python -m pytest tests/test_sampler_gpu.py -q -m gpu 2>&1 | tail -2) > $O/pytest_inflight1.txt
grep -h "induce_kernel" $O/kernel_stats_*.csv | cut -c1-40,150-260
