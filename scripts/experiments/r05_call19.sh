#!/bin/bash
# round 5, call 19: has anything since call 10 cost the config-4 kernels (certificate bound, sector-certificate branch)?
export TMPDIR=/tmp
for dt in f64 f32; do for dm in mindist slicer; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype $dt --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $dt $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
timeout 600 python -m pytest tests/test_demod_cert.py tests/test_gpu_tdl_wave.py -m gpu -q --timeout=600 2>&1 | tail -3
