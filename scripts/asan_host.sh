#!/bin/bash
# AddressSanitizer run of the HOST side of the C ABI: builds asan/libmcle_asan.so
# (`make asan`, device code not instrumented) and runs the CPU-reachable tests of the boundary on it -- the library loads, every
# declared symbol resolves, argument errors are reported, contexts / options / error strings round-trip -- with leak detection off
# (CPython and the HIP runtime keep allocations until exit) and everything else on.  No GPU needed: the compute entry points fail
# with "no device" after their argument checks, which is the part of the host code a CPU box reaches.
# usage: bash scripts/asan_host.sh [log]      (from the repo root; ~4 min for the build the first time)
set -e
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06/asan_host.log}
SAN="-fsanitize=address -fno-gpu-sanitize -shared-libsan"
make -C pyphysim_amd/csrc -j8 asan SAN="$SAN" > /tmp/asan_build.log 2>&1
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
{
  echo "# scripts/asan_host.sh: $(date -u +%F) host AddressSanitizer build of libmcle (hipcc, host code only, shared sanitizer runtime)"
  echo "# runtime: $RT"
  MCLE_LIBRARY=$PWD/pyphysim_amd/csrc/asan/libmcle_asan.so LD_PRELOAD=$RT \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_odr_violation=0:protect_shadow_gap=0 \
    python -m pytest tests/test_capi_library.py tests/test_capi_errors_cpu.py tests/test_demod_grid.py tests/test_demod_cert.py tests/test_host_mirror_cpu.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} | tee "$LOG"
