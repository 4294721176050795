#!/bin/bash
# round 5, call 9: product build -- every test that touches the frequency-selective MIMO link, then the rates
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_simulators.py tests/test_gpu_legacy.py tests/test_gpu_fuzz.py -m gpu -q --timeout=900 2>&1 | tail -8
for dt in f32 f64; do for k in 0 1; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $dt --config f1 --batch 98304 --opt mimo_tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 $dt kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
