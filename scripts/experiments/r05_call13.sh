#!/bin/bash
# round 5, call 13: the symbol walks of config 5 and f6 -- complex128 receiver by receiver at three wavefronts per SIMD, complex64
# 4-PSK through the quadrant certificate + explicit packed forms: parity, then rates (round-4 box: c5 1.24e8 / 2.22e8, f6 5.1e7 / 1.46e8)
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bd.py tests/test_gpu_oracle_depth.py tests/test_gpu_pipelines.py -m gpu -q --timeout=900 -k "bd or ia or config5 or block_diag" 2>&1 | tail -5
for cfg in c5 f6; do for dt in f64 f32; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $dt --config $cfg --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $dt', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
