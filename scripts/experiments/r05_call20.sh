#!/bin/bash
# round 5, call 21: sector certificate inside the grid searches (A = D = as committed), E = the same without the QAM certificate's magnitude bound
# the config-4 kernels; A = the library as committed
export TMPDIR=/tmp
cp pyphysim_amd/csrc/libmcle.so /tmp/libA.so
for rep in 1 2; do for which in A D E; do
  if [ $which = A ]; then cp /tmp/libA.so pyphysim_amd/csrc/libmcle.so; else cp gpurun_in/lib$which.so pyphysim_amd/csrc/libmcle.so; fi
  for dt in f64 f32; do
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist --dtype $dt --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which c4 $dt mindist', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
  done
done; done
cp /tmp/libA.so pyphysim_amd/csrc/libmcle.so
