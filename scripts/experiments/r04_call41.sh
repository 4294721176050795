#!/bin/bash
# round 4, call 41: wavefront kernel -- polynomial orders up to 8, records up to 128 values, the envelope's corners; rates unchanged?
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tdl_wave.py tests/test_gpu_pipelines.py tests/test_gpu_slices.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert" | tail -10
for dt in f32 f64; do
  python bench.py --steps 10 --warmup 3 --no-cpu --pmc off --dtype $dt --config c3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $dt', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'frac %.3f' % d['roofline']['frac'])"
done
