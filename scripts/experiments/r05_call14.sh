#!/bin/bash
# round 5, call 14: A/B of the config-5 / f6 walks: the library of the previous commit (gpurun_in/libmcle_old.so) against this one,
# alternating, same box, both demodulators of config 5
export TMPDIR=/tmp
cp pyphysim_amd/csrc/libmcle.so /tmp/libmcle_new.so
for rep in 1 2; do for which in old new; do
  if [ $which = old ]; then cp gpurun_in/libmcle_old.so pyphysim_amd/csrc/libmcle.so; else cp /tmp/libmcle_new.so pyphysim_amd/csrc/libmcle.so; fi
  for spec in "c5 f64 mindist" "c5 f64 slicer" "c5 f32 mindist" "c5 f32 slicer" "f6 f64 mindist" "f6 f32 mindist"; do
    set -- $spec
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $2 --demod $3 --config $1 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which $spec', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
  done
done; done
cp /tmp/libmcle_new.so pyphysim_amd/csrc/libmcle.so
