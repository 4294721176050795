#!/bin/bash
# round 5, call 24: two A/Bs of whole library builds on one box (MCLE_LIBRARY picks the build):
#  (1) f1's frequency response: libmcle_prev.so (mean x twiddle once, then u0 += t, u1 += (-1)^d t: five instructions per entry and
#      tap) against libmcle.so (one complex multiply-add into the tap's delay class, a butterfly behind the loop: two per entry and tap)
#  (2) the complex64 symbol walks (k_ia_link, k_bd_link, k_mimo_flat_link) bounded for four wavefronts per SIMD (128 registers, 2 - 78
#      spilled) against three (168 registers, nothing spilled): libmcle.so against libmcle_lb3.so (-DMCLE_F32_WALK_WAVES=3)
export TMPDIR=/tmp
L=$PWD/pyphysim_amd/csrc
one() { lib=$1; tag=$2; shift 2
  MCLE_LIBRARY=$L/$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $tag', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"; }
for round in 1 2; do
  for lib in libmcle_prev.so libmcle.so; do
    one $lib "f1 f32" --config f1 --dtype f32 --demod slicer --batch 98304
    one $lib "f1 f64" --config f1 --dtype f64 --batch 98304
  done
  for lib in libmcle.so libmcle_lb3.so; do
    one $lib "c5 f32 slicer" --config c5 --dtype f32 --demod slicer --batch 1048576
    one $lib "c5 f32 mindist" --config c5 --dtype f32 --demod mindist --batch 1048576
    one $lib "f6 f32" --config f6 --dtype f32 --batch 524288
    MCLE_LIBRARY=$L/$lib timeout 300 python scripts/bench_mimo_schemes.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib flat MIMO', ' '.join('%s %.3f ms' % (k, v['kernel_ms']) for k,v in d.items()))"
  done
done
# other f1 shapes on the new build, against the numbers of the previous one
for lib in libmcle_prev.so libmcle.so; do MCLE_LIBRARY=$L/$lib timeout 300 python scripts/experiments/r05_f1_shapes.py $lib; done
