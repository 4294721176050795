#!/bin/bash
# round 5, call 3: stage ablation of the complex64 wavefront f1 kernel (MCLE_EXPERIMENTS build) + register / work-item shapes
export TMPDIR=/tmp
for k in 0 1 17 18 20 24 32 47 56 57 58; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f32 --config f1 --batch 83886 --opt mimo_tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 f32 kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
