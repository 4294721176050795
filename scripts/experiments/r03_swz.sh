#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_mfma.py tests/test_gpu_fuzz.py tests/test_gpu_operators.py -m gpu -q --timeout=900 -x 2>&1 | tail -2
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --demod slicer "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config f1 --dtype f32 --batch 98304
run --config f1 --dtype f64 --batch 98304
run --config c3 --dtype f64 --batch 131072
run --config c3 --dtype f32 --batch 131072 --opt no_mfma=1
run --config c4 --dtype f32 --batch 65536 --opt no_mfma=1
run --config c4 --dtype f64 --batch 65536 --opt f64_generic=1
