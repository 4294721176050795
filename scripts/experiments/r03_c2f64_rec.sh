#!/bin/bash
# config 2 complex128: ray rotation recurrence over a thread's 16 symbols vs one sincos per ray and sample
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_slices.py tests/test_gpu_simulators.py -m gpu -q --timeout=900 -x 2>&1 | grep -E "passed|failed|FAILED|error" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config c2 --dtype f64 --batch 16384 --demod slicer
run --config c2 --dtype f64 --batch 16384 --demod mindist
run --config c2 --dtype f64 --batch 16384 --demod mindist --opt jakes_direct=1
run --config c3 --dtype f64 --batch 131072 --demod mindist
