#!/bin/bash
# round 4, call 6: the family kernel (1024, 4x4) against the benchmark-shaped kernel it generalised, same box, alternating
export TMPDIR=/tmp
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
cp pyphysim_amd/csrc/libmcle.so /tmp/new.so
for rep in 1 2 3; do
  cp /tmp/new.so pyphysim_amd/csrc/libmcle.so; one "family  mindist"; one "family  slicer" --demod slicer
  cp gpurun_in/libmcle_old.so pyphysim_amd/csrc/libmcle.so; one "special mindist"; one "special slicer" --demod slicer
done
cp /tmp/new.so pyphysim_amd/csrc/libmcle.so
