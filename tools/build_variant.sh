#!/bin/bash
# Timing experiments: build the library with extra -D flags into gcc_amd/csrc/variants/lib_<name>.so (git-ignored; it travels to
# the GPU box with the snapshot, unlike gpurun_out/).  On the box a script swaps it in for the run only:
#     cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so
#     cp gcc_amd/csrc/variants/lib_<name>.so gcc_amd/csrc/libgcc_amd.so;  python bench.py ...;  cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
# or passes it to tools/graph_probe.py --lib.  The product never loads anything but gcc_amd/csrc/libgcc_amd.so (_cabi.load).
#   tools/build_variant.sh edge_fill -DGCC_POSEMB_EDGE_FILL=1
#   tools/build_variant.sh j4 -DGIN_GATHER_J=4 -DGIN_IN_PER_CU=4
# Knobs: GIN_GATHER_J, GIN_IN_PER_CU, GIN_IN_LDS_W, GIN_DBG_SKIP, GATHER_DBG, GCC_KREP (statistics replicas: 8 / 16) (encoder.hip / encoder_common.h); BWD_GATHER_J, BWD_C_PER_CU
# (encoder_bwd.hip); NCE_DQ_BATCH, GCC_NCE_WGS_DEFAULT (nce.hip; run time: GCC_NCE_WGS); GCC_POSEMB_MID_THREADS, GCC_POSEMB_EDGE_FILL, GCC_POSEMB_W48_OCC, GCC_POSEMB_CH_LONGDEG, GCC_POSEMB_CH_NARROW_WANT, GCC_POSEMB_ABLATE_MID / _CHEB / _WAVE (timing only: that eigensolver class returns at once) (posemb.hip); GCC_GINW_ABLATE (gin_wide.hip).
set -eu
name=$1; shift
cd "$(dirname "$0")/../gcc_amd/csrc"
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" -o variants/lib_$name.so \
    common.hip sampler.hip encoder.hip encoder_bwd.hip encoder_eval.hip nce.hip posemb.hip gin_wide.hip ginx.hip
ls -la variants/lib_$name.so
