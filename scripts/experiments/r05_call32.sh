#!/bin/bash
# round 5, call 32: config 2 in complex128 with the ray sum as run-start phasors x a table of powers (one chained multiply-add per
# ray instead of a rotation and two additions) -- libmcle.so against the previous build; then every suite that runs the flat kernels
export TMPDIR=/tmp
mkdir -p gpurun_out
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "c2 f64 mindist" --config c2 --dtype f64 --batch 16384
    one $lib "c2 f64 slicer" --config c2 --dtype f64 --demod slicer --batch 16384
  done
done
timeout 1800 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py tests/test_gpu_simulators.py tests/test_gpu_slices.py -q --timeout=900 > gpurun_out/pytest_c2.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_c2.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/pytest_c2.log | head
