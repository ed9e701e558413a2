#!/bin/bash
# GPU tests of the host layer against the AddressSanitizer + UBSan build of the library (make -C zaf-python_amd/csrc asan):
#   gpurun -- 'bash tools/asan_run.sh [pytest -k expression]'
# The interpreter is not instrumented, so the sanitizer runtime is preloaded; leaks are not reported (CPython and the HIP runtime keep
# theirs).
cd "$(dirname "$0")/.." || exit 1
RT=$(gcc -print-file-name=libasan.so)   # GCC's runtime (ROCm's intercepts the HSA allocator: see the Makefile)
K=${1:-"run_host or concurrent or pcm or chunks or placed or one_pass or host"}
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT ZAFX_LIBRARY=$PWD/tools/bin/libzafx_asan.so python -m pytest tests -x -q -m gpu -k "$K" -p no:cacheprovider
