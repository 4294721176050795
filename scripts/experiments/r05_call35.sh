#!/bin/bash
# round 5, call 35: config 3 complex64 with the delay line's Horner + multiply-add as four packed instructions per tap and sample
# (chan_step, pkcx.hpp: the form f1 uses) instead of eight scalar FMAs -- libmcle.so against the previous build; then its suites
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c3 f32 262144" --config c3 --dtype f32 --batch 262144
    one $lib "c3 f32 2^21" --config c3 --dtype f32 --batch 2097152
  done
done
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 300 python scripts/bench_tdl_family.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', {k: '%.4g' % v['wave']['realizations_per_s'] if isinstance(v, dict) and 'wave' in v else v for k, v in d.items() if 'f32' in k})"; done
timeout 1200 python -m pytest tests/test_gpu_tdl_wave.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py -q --timeout=900 2>&1 | tail -3
