#!/bin/bash
# complex128 grid search: first four candidates straight-line (table entries fetched together)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_pipelines.py tests/test_gpu_operators.py tests/test_gpu_fuzz.py tests/test_gpu_staged_c4.py -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config c4 --dtype f64 --batch 262144 --demod mindist
run --config f1 --dtype f64 --batch 98304 --demod mindist
run --config c5 --dtype f64 --batch 262144 --demod mindist
run --config f6 --dtype f64 --batch 131072 --demod mindist
