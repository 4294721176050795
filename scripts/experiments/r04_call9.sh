#!/bin/bash
# round 4, call 9: radix-16 kernel with the fused middle stage (pass C + channel + pass C') -- parity, then A/B
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=900 2>&1 | tail -8
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
for rep in 1 2; do
one "radix-16 fused   " --demod mindist
one "radix-16 unfused " --demod mindist --opt f64_threads=257
one "radix-4  512thr  " --demod mindist --opt f64_threads=512
one "radix-16 fused slicer" --demod slicer
done
