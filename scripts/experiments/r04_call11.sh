#!/bin/bash
# round 4, call 11: radix-16 kernel with every layer-1 twiddle from the table (default) against register x 16th root (258)
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=900 2>&1 | tail -5
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
for rep in 1 2 3; do
one "r16 fused exact-tw " --demod mindist
one "r16 fused root-tw  " --demod mindist --opt f64_threads=258
done
one "r16 fused exact slicer" --demod slicer
