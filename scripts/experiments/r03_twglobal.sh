#!/bin/bash
# complex128 TDL kernels with the twiddle table read from global instead of LDS (more workgroups per CU)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_slices.py -m gpu -q --timeout=900 -x 2>&1 | grep -E "passed|failed|FAILED|error" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config f1 --dtype f64 --batch 98304 --demod slicer
run --config f1 --dtype f64 --batch 98304 --demod mindist
run --config c3 --dtype f64 --batch 131072 --demod slicer
run --config c3 --dtype f64 --batch 131072 --demod mindist
run --config f1 --dtype f32 --batch 98304 --demod slicer
