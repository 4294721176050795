#!/bin/bash
# round 5, call 36: complex64 -- the planar config-4 kernels' H x and G y multiply-adds and config 3's equaliser taps as two packed
# FMAs (pk_cfma) instead of four scalar ones: libmcle.so against the previous build; then the suites
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c4 f32 mindist" --config c4 --dtype f32 --demod mindist --batch 262144
    one $lib "c4 f32 slicer" --config c4 --dtype f32 --demod slicer --batch 262144
    one $lib "c3 f32 2^21" --config c3 --dtype f32 --batch 2097152
  done
done
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 400 python scripts/bench_f64_family.py --dtype f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', {k: '%.4g' % v['fast_mindist']['realizations_per_s'] for k, v in d.items() if isinstance(v, dict) and 'fast_mindist' in v})"; done
timeout 1500 python -m pytest tests/test_gpu_planar_f32.py tests/test_gpu_tdl_wave.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py tests/test_gpu_pipelines.py tests/test_gpu_f64_kernel.py -q --timeout=900 2>&1 | tail -3
