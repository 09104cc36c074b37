#!/bin/bash
# Host-side sanitizer run (CPU container only -- GPU AddressSanitizer is not available on this pool and is not what this is):
# builds build/asan/librvcx_asan.so (make host-asan) and runs tools/host_asan_driver.py on it with the ASan runtime preloaded.
set -e
cd "$(dirname "$0")/.."
make -j8 host-asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
# python itself is not leak-clean; everything else is fatal
exec env LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 RVCX_LIBRARY="$PWD/build/asan/librvcx_asan.so" RVCX_DEBUG=1 \
    python3 tools/host_asan_driver.py
