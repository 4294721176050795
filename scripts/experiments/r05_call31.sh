#!/bin/bash
# round 5, call 31: complex128 symbol loops without IEEE division / square-root sequences (rcp_newton, sqrt_rsqrt_newton in
# common.hpp): config 2's equaliser as s + z conj(h) / |h|^2, config 3's one-tap equaliser, f1's Cholesky pivots -- libmcle.so
# against the previous build; then the suites that cover them
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c2 f64" --config c2 --dtype f64 --batch 16384
    one $lib "c3 f64" --config c3 --dtype f64 --batch 131072
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
  done
done
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
