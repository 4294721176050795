#!/bin/bash
# round 4, call 31: config 3 with one realization per wavefront (option tdl_wave=1): parity, then rates against the batched kernels
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tdl_wave.py -m gpu -q --timeout=600 -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert|Error" | tail -15
for dt in f32 f64; do for w in 0 1; do
  b=262144; [ $dt = f64 ] && b=131072
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --dtype $dt --config c3 --batch $b --opt tdl_wave=$w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $dt tdl_wave=$w', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
timeout 600 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_planar_f32.py -m gpu -q --timeout=600 -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -4
