#!/bin/bash
# Round 5, call 20: eval kernels with the weight rows fetched in a per-workgroup rotated order (512 workgroups no longer queue up on the
# same lines) against the same order everywhere (variant build -DGCC_EVAL_ROWROT=0): eval probe at rw_hops 64 and 256, eval parity tests.
set -u
O=gpurun_out/r5c20
mkdir -p $O
export TMPDIR=/tmp GCC_AMD_GRAPH_CACHE=/tmp/graphs
timeout 900 python -m pytest tests/test_generate_gpu.py tests/test_encoder_gpu.py tests/test_hidden_size_gpu.py -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1
echo "== tests: $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1)"; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
probe() { (timeout 300 python tools/eval_probe.py "$@" 2>&1 | tail -3) | cut -c1-700; }
echo "-- rotated"; probe > $O/eval_probe_rot.txt; cat $O/eval_probe_rot.txt
probe --rw-hops 256 > $O/eval_probe_rot_256.txt 2>&1; cat $O/eval_probe_rot_256.txt
cp gcc_amd/csrc/libgcc_amd.so /tmp/lib_default.so; cp gcc_amd/csrc/variants/lib_norot.so gcc_amd/csrc/libgcc_amd.so
echo "-- same order everywhere"; probe > $O/eval_probe_norot.txt; cat $O/eval_probe_norot.txt
probe --rw-hops 256 > $O/eval_probe_norot_256.txt 2>&1; cat $O/eval_probe_norot_256.txt
cp /tmp/lib_default.so gcc_amd/csrc/libgcc_amd.so
