#!/bin/bash
# round 4, call 8: the radix-16 variant of the (1024, 4x4) complex128 kernel -- parity, then time against the two radix-4 forms
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_operators.py tests/test_gpu_staged_c4.py -m gpu -q --timeout=900 -x 2>&1 | tail -8
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
for rep in 1 2; do
one "512thr radix-4 " --demod mindist
one "256thr radix-4 " --demod mindist --opt f64_threads=256
one "256thr radix-16" --demod mindist --opt f64_variant=4
one "256thr radix-16 slicer" --demod slicer --opt f64_variant=4
done
