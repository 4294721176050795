#!/bin/bash
# round 4, call 2: whole GPU suite on the certificate + v_bitop3 Philox build; config-4 complex128: plain kernel and the
# timing bound of the channel fusion (f64_variant=32: wrong results by construction).
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -12
one() {
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
for dm in mindist slicer; do
  one "512thr $dm var0" --demod $dm
  one "512thr $dm var32(bound)" --demod $dm --opt f64_variant=32
  one "256thr $dm var0" --demod $dm --opt f64_threads=256
done
one "c4 f32 mindist" --dtype f32 --demod mindist
one "c4 f32 slicer" --dtype f32 --demod slicer
