#!/bin/bash
# round 5, call 15: config 2 in complex128 -- one reciprocal in the equaliser, exact phasor restart per 64 symbols in full chunks:
# parity at full size, then old library against new
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_oracle_depth.py -m gpu -q --timeout=900 -k "flat or config2" 2>&1 | tail -4
cp pyphysim_amd/csrc/libmcle.so /tmp/libmcle_new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp gpurun_in/libmcle_old.so pyphysim_amd/csrc/libmcle.so; else cp /tmp/libmcle_new.so pyphysim_amd/csrc/libmcle.so; fi
  for spec in "c2 f64 mindist" "c2 f64 slicer"; do
    set -- $spec
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype $2 --demod $3 --config $1 --batch 16384 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $spec', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"
  done
done; done
cp /tmp/libmcle_new.so pyphysim_amd/csrc/libmcle.so
