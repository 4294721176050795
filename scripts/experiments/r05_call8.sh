#!/bin/bash
# round 5, call 8: the channel stage without its LDS reads (timing bound), the default back at three wavefronts per SIMD
export TMPDIR=/tmp
for k in 0 80 17 1; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f32 --config f1 --batch 83886 --opt mimo_tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 f32 kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
