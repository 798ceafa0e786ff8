#!/bin/bash
# The emulator tier under AddressSanitizer: the kernels' accesses to workspace / batch buffers (numpy / torch-CPU heap memory) are
# checked for out-of-bounds reads and writes.  (LDS is one static buffer of the emulator: an overrun inside it is not seen.)
# Round 4: sampler, eval, train-step, hidden-size, wide-GIN, NCE, pos-emb, encoder and headline-step emulator tests all clean.
#   scripts/asan_emu.sh [pytest args ...]            default: tests/test_sampler_emu.py
set -eu
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
touch gcc_amd/csrc/sampler.hip
make -C gcc_amd/csrc emu EXTRA="-fsanitize=address -fno-omit-frame-pointer" > /dev/null
rc=0
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$ASAN python -m pytest "${@:-tests/test_sampler_emu.py}" -x -q || rc=$?
touch gcc_amd/csrc/sampler.hip            # back to the plain build: the sanitised library does not load without the preload
make -C gcc_amd/csrc emu > /dev/null
exit $rc
