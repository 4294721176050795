#!/bin/bash
# round 4, call 22: complex64 certificate margin 2^-15 (was 2^-12): whole suite, then the complex64 min-distance rates
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
for cfg in c4 c2 c5 f1; do
  b=262144; [ $cfg = c2 ] && b=65536; [ $cfg = f1 ] && b=98304; [ $cfg = c5 ] && b=1048576
  for dm in mindist slicer; do
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype f32 --config $cfg --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg f32 $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
  done
done
