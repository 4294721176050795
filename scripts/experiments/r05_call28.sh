#!/bin/bash
# round 5, call 28: acc + a b as four chained FMAs in the complex128 fused pipelines (cfma4, common.hpp) against the generic
# product-then-add form (mul + fma + add per component): libmcle.so against libmcle_prev.so, two alternating rounds; then the suites
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c4 f64 mindist" --config c4 --dtype f64 --demod mindist --batch 262144
    one $lib "c3 f64" --config c3 --dtype f64 --batch 131072
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
    one $lib "c5 f64 mindist" --config c5 --dtype f64 --demod mindist --batch 262144
    one $lib "f6 f64" --config f6 --dtype f64 --batch 131072
  done
done
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -5
