#!/bin/bash
# complex128 Box-Muller tables in LDS for the generic kernels (C2, C3, C5, f6): timings + the parity suites
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_bd.py tests/test_gpu_ia_base.py tests/test_gpu_simulators.py -m gpu -q --timeout=900 -x 2>&1 | grep -E "passed|failed|FAILED|error" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config c3 --dtype f64 --batch 131072
run --config c5 --dtype f64 --batch 262144
run --config f6 --dtype f64 --batch 131072
run --config c5 --dtype f32 --batch 262144
run --config f6 --dtype f32 --batch 131072
