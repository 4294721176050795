#!/bin/bash
# round 5, call 27: f1's H(f) by delay-class POSITIONS (host-sorted: even-delay taps at positions 0.., odd-delay ones at 7, 6, ..; two
# straight loops, one complex multiply-add per entry and tap, no per-tap class branch) -- libmcle.so -- against the sign form of
# the previous build (libmcle_prev.so), then the f1 suite on the new build
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "f1 f32" --config f1 --dtype f32 --demod slicer --batch 98304
    one $lib "f1 f32 mindist" --config f1 --dtype f32 --demod mindist --batch 98304
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
  done
done
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 300 python scripts/experiments/r05_f1_shapes.py $lib; done
timeout 1500 python -m pytest tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_fuzz.py tests/test_gpu_simulators.py -q --timeout=900 2>&1 | tail -4
