#!/bin/bash
# the lane emulator (the device code compiled for the CPU, tests/emu) under AddressSanitizer + UBSan: out-of-bounds reads of the
# model tables and of the LDS block, misaligned accesses, signed overflow in index arithmetic.  GPU sanitizers are not available
# on the pool; this is the CPU stand-in.   usage: bash tools/emu_asan.sh [pytest -k expression]
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/librkfd_emu_asan.so
g++ -std=c++20 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -fPIC -shared -pthread \
    -Iinclude -Iroki-fd_amd/csrc -Iroki-fd_amd/csrc/host -Iroki-fd_amd/build -o $OUT tests/emu/rkfd_emu.cpp roki-fd_amd/csrc/rkfd_devmodel.cpp
# the same harness with two instances per wavefront (RKFD_W = 2)
g++ -std=c++20 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -fPIC -shared -pthread -DRKFD_W=2 \
    -Iinclude -Iroki-fd_amd/csrc -Iroki-fd_amd/csrc/host -Iroki-fd_amd/build -o /tmp/librkfd_emu_asan_w2.so tests/emu/rkfd_emu.cpp roki-fd_amd/csrc/rkfd_devmodel.cpp
export RKFD_EMU_LIB=$OUT
export RKFD_EMU_LIB_W2=/tmp/librkfd_emu_asan_w2.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" python3 -m pytest tests/test_emu_parity.py -x -q -k "${1:-config}"
