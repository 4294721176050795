#!/bin/bash
# round 4, call 48: grid rules of call 46 / 47 applied (wavefront kernels: >= 16 units per workgroup; planar kernels: up to 16 x resident)
export TMPDIR=/tmp
python scripts/experiments/r04_oversub_probe.py 2>&1 | grep "oversub 0"
for cfg in "c4 f64 262144" "c4 f32 262144" "c4 f64 1048576" "c3 f32 2097152" "c3 f64 2097152" "c3 f32 131072" "c3 f64 131072"; do
  set -- $cfg
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist --dtype $2 --config $1 --batch $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 batch $3', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
done
timeout 600 python -m pytest tests/test_gpu_tdl_wave.py tests/test_gpu_f64_kernel.py tests/test_gpu_planar_f32.py tests/test_gpu_slices.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed" | tail -2
