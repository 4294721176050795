#!/bin/bash
# round 5, call 5: explicit packed complex ops (pkcx.hpp) in the channel, H(f) and the Gram rows; complex64 at two wavefronts per SIMD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mimo_tdl_wave.py -m gpu -q --timeout=600 -x -k "against_the_oracle and not 240 and not 120" 2>&1 | tail -5
for k in 0 57 1 17 18 24 47; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f32 --config f1 --batch 83886 --opt mimo_tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 f32 kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config f1 --batch 29127 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 f64', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
