#!/bin/bash
# round 4, call 7: whole GPU suite on the current build, family (1024, 4x4) vs the benchmark-shaped kernel again, then the
# round's rocprofv3 profiles of the min-distance kernels (LDS conflict fractions with the certificate) and the headline
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -6
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
cp pyphysim_amd/csrc/libmcle.so /tmp/new.so
for rep in 1 2; do
  cp /tmp/new.so pyphysim_amd/csrc/libmcle.so; one "family  mindist"
  cp gpurun_in/libmcle_old.so pyphysim_amd/csrc/libmcle.so; one "special mindist"
done
cp /tmp/new.so pyphysim_amd/csrc/libmcle.so
bash scripts/prof_r04.sh c4_f64 c4_f64sl c4md c4 c2_f64 c5_f64 c2 c5 2>&1 | tail -8
