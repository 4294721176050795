#!/bin/bash
# round 5, call 40: the complex64 forms of the benchmark geometry again now that the multiply-adds are packed and nothing spills:
# f64_threads = 0 (default: radix-16 passes, separate channel stage, four wavefronts per SIMD), 259 (fused channel stage), 257 (three
# wavefronts per SIMD), two rounds
export TMPDIR=/tmp
for round in 1 2; do for thr in 0 259 257; do for dm in mindist slicer; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --config c4 --dtype f32 --demod $dm --batch 262144 --opt f64_threads=$thr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f64_threads=$thr $dm', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.7f' % d['ser'])"
done; done; done
