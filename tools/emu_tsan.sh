#!/bin/bash
# the lane emulator (64 threads = 64 lanes, SYNC() = a barrier; tests/emu) under ThreadSanitizer: two lanes touching the same LDS
# or global address without a barrier between them.  On the GPU a wavefront's lanes run in lockstep, but the compiler only keeps
# LDS accesses ordered across lanes where the code says so; a race here is a missing SYNC() there.
# usage: bash tools/emu_tsan.sh [pytest -k expression]     reports: /tmp/rkfd_tsan.<pid>
# (the last test's oracle teardown segfaults under the TSan runtime only - not under ASan, not natively; the device code's
# reports are complete by then)
# (validated by dropping a quarter of the barriers: 23 reports and a failing test; the code as committed: none)
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/librkfd_emu_tsan.so
g++ -std=c++20 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Wno-unknown-pragmas -fPIC -shared -pthread \
    -Iinclude -Iroki-fd_amd/csrc -Iroki-fd_amd/csrc/host -Iroki-fd_amd/build -o $OUT tests/emu/rkfd_emu.cpp roki-fd_amd/csrc/rkfd_devmodel.cpp
rm -f /tmp/rkfd_tsan.*
# the same harness with two instances per wavefront (RKFD_W = 2)
g++ -std=c++20 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Wno-unknown-pragmas -fPIC -shared -pthread -DRKFD_W=2 \
    -Iinclude -Iroki-fd_amd/csrc -Iroki-fd_amd/csrc/host -Iroki-fd_amd/build -o /tmp/librkfd_emu_tsan_w2.so tests/emu/rkfd_emu.cpp roki-fd_amd/csrc/rkfd_devmodel.cpp
export RKFD_EMU_LIB=$OUT
export RKFD_EMU_LIB_W2=/tmp/librkfd_emu_tsan_w2.so
export TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:log_path=/tmp/rkfd_tsan"
LD_PRELOAD="$(g++ -print-file-name=libtsan.so)" python3 -m pytest tests/test_emu_parity.py -x -q -k "${1:-emulated}" || true
echo "ThreadSanitizer reports: $(cat /tmp/rkfd_tsan.* 2>/dev/null | grep -c 'WARNING: ThreadSanitizer')"
