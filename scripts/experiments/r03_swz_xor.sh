#!/bin/bash
# radix-4 swizzles as one swizzle + three XORs with per-stage constants (XOR-linearity of lds_swz)
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_operators.py -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -3
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config c4 --dtype f64 --batch 262144 --demod mindist
run --config c4 --dtype f64 --batch 262144 --demod slicer
run --config f1 --dtype f32 --batch 98304 --demod slicer
run --config f1 --dtype f64 --batch 98304 --demod mindist
run --config c3 --dtype f64 --batch 131072
run --config c4 --dtype f32 --batch 65536 --opt no_mfma=1 --demod slicer
run --config c4 --dtype f64 --batch 65536 --opt f64_generic=1 --demod slicer
