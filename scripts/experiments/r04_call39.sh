#!/bin/bash
# round 4, call 39: branch-free decode pass in the 256-thread planar config-4 kernels: parity, rates (both arithmetics, both demodulators)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_planar_f32.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -4
for dt in f64 f32; do for dm in mindist slicer; do
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype $dt --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $dt $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
