export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
for i in 1 2; do timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^E " | head -8 | cut -c1-250; done
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_before_longfirst.so gcc_amd/csrc/libgcc_amd.so
echo "-- before long-first"
for i in 1 2; do timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -E "passed|failed|^E " | head -8 | cut -c1-250; done
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
