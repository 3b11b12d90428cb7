#!/bin/bash
# The CPU wave-machine tests (tests/hostsim) with the kernel sources built under AddressSanitizer + UBSan-free plain ASan: every
# out-of-bounds access of a kernel into a caller's tensor, a static LDS array or a stack array stops the run with the kernel's
# file:line.  Usage: tools/hostsim_asan.sh [pytest args]   (default: tests/test_hostsim_cpu.py)
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then set -- tests/test_hostsim_cpu.py; fi
HOSTSIM_FULL=1 HOSTSIM_FLAGS="-fsanitize=address -shared-libasan -fno-omit-frame-pointer -g" LD_PRELOAD=$RT \
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0 \
python -m pytest "$@" -q -x -p no:cacheprovider
