#!/bin/bash
# round 5, call 39: f1's fading pre-kernel with the PHASE draws shared inside quads of lanes (one Philox block per lane and ray pair
# instead of four) -- libmcle.so against the previous build, both arithmetics; then the f1 suites (complex128: exact counts)
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "f1 f32" --config f1 --dtype f32 --demod slicer --batch 98304
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
  done
done
timeout 1200 python -m pytest tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_fuzz.py tests/test_gpu_simulators.py tests/test_gpu_pipelines.py tests/test_gpu_legacy.py -q --timeout=900 2>&1 | tail -3
