#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config f1 --dtype f32 --batch 98304 --demod mindist
run --config f1 --dtype f32 --batch 98304 --demod slicer
