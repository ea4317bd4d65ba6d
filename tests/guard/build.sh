#!/bin/bash
# Builds the red-zone guard allocator (test infrastructure; see guard_allocator.cpp) into tests/guard/_build/libgvd_guard.so
set -e
H=$(cd "$(dirname "$0")" && pwd)
mkdir -p $H/_build
${HIPCC:-/opt/rocm/bin/hipcc} -O2 -std=c++17 -fPIC -shared -Wall -x c++ -D__HIP_PLATFORM_AMD__=1 -I${ROCM_PATH:-/opt/rocm}/include \
    -o $H/_build/libgvd_guard.so $H/guard_allocator.cpp -L${ROCM_PATH:-/opt/rocm}/lib -lamdhip64 -Wl,-rpath,${ROCM_PATH:-/opt/rocm}/lib
