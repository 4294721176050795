#!/bin/bash
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | grep -E "passed|failed|FAILED" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --demod slicer "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config f1 --dtype f32 --batch 98304
run --config f1 --dtype f64 --batch 98304
run --config c3 --dtype f64 --batch 131072
run --config c3 --dtype f32 --batch 131072 --opt no_mfma=1
run --config c4 --dtype f32 --batch 65536 --opt no_mfma=1
run --config c4 --dtype f64 --batch 65536 --opt f64_generic=1
run --config c4 --dtype f64 --batch 262144
run --config c4 --dtype f64 --batch 262144 --demod mindist
