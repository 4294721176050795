#!/bin/bash
# round 4, call 34: the quadrant certificate on / off (demod_nocert) in config 3, both kernels, both arithmetics
export TMPDIR=/tmp
for dt in f32 f64; do for w in 0 1; do for nc in 0 1; do
  b=262144; [ $dt = f64 ] && b=131072
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --dtype $dt --config c3 --batch $b --opt tdl_wave=$w --opt demod_nocert=$nc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $dt tdl_wave=$w nocert=$nc', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done; done
