#!/bin/bash
# round 5, call 48: the per-lane pick of a Philox block's words (cn_sample, symbol_at, uniform_at: philox.hpp) as selects instead of
# b.w[index] -- the lane-dependent index had the backend park the four words in SCRATCH (a 16-byte store + a dependent load per
# sample: nine round trips per symbol column of the config-5 walk) -- libmcle.so against the previous build; then the suites
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c5 f64 mindist" --config c5 --dtype f64 --demod mindist --batch 262144
    one $lib "c5 f32 slicer" --config c5 --dtype f32 --demod slicer --batch 1048576
    one $lib "f6 f64" --config f6 --dtype f64 --batch 131072
    one $lib "f6 f32" --config f6 --dtype f32 --batch 524288
  done
done
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 300 python scripts/bench_mimo_schemes.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib flat MIMO f32', ' '.join('%s %.3f ms' % (k, v['kernel_ms']) for k,v in d.items()))"; done
timeout 1500 python -m pytest tests/test_gpu_ia_base.py tests/test_gpu_bd.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py tests/test_gpu_simulators.py tests/test_gpu_pipelines.py -q --timeout=900 2>&1 | tail -3
