#!/bin/bash
# round 5, call 34: f1 complex128 with the fading pre-kernel's per-ray sincos as the polynomial form (bm_sincos_rad) instead of the
# library routine -- libmcle.so against the previous build; then the f1 suites
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
  done
done
timeout 1200 python -m pytest tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py tests/test_gpu_simulators.py -q --timeout=900 2>&1 | tail -3
