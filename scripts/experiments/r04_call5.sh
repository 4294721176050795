#!/bin/bash
# round 4, call 5: which part of the channel-fusion bound is barriers, which is LDS stores (f64_variant 1 / 2 / 3: wrong results)
export TMPDIR=/tmp
one() {
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
one "plain" --demod mindist
one "no stores (1)" --demod mindist --opt f64_variant=1
one "no barriers (2)" --demod mindist --opt f64_variant=2
one "neither (3)" --demod mindist --opt f64_variant=3
one "plain again" --demod mindist
one "256thr" --demod mindist --opt f64_threads=256
