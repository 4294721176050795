#!/bin/bash
# round 5, call 29: f1 complex128 with the Gram rows and the Cholesky solve as chained FMAs (libmcle.so) against call 28's build
# (libmcle_prev.so), then the whole GPU suite (log kept)
export TMPDIR=/tmp
mkdir -p gpurun_out
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
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 300 python scripts/experiments/r05_f1_shapes.py $lib | grep f64; done
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3
scripts/experiments/bin/hf_mfma_ab
