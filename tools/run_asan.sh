#!/bin/bash
# CPU container only: sanitizer build of libapgp.so's host side + the C-ABI checks under it (tools/asan_cabi.py).
set -e
cd "$(dirname "$0")/.."
make -C approxposterior_amd/csrc asan -j4
RT=/usr/lib/x86_64-linux-gnu
LD_PRELOAD=$RT/libasan.so.6:$RT/libubsan.so.1 \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python3 tools/asan_cabi.py
