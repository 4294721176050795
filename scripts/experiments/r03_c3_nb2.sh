#!/bin/bash
# config 3 matrix-core kernel: 3 waves per SIMD with two realizations per pass (option tdl_mfma_waves=32) vs the default
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mfma.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -5
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"; }
run --config c3 --dtype f32 --batch 131072
run --config c3 --dtype f32 --batch 131072 --opt tdl_mfma_waves=32
run --config c3 --dtype f32 --batch 131072 --opt tdl_mfma_waves=3
run --config c3 --dtype f32 --batch 262144
run --config c3 --dtype f32 --batch 262144 --opt tdl_mfma_waves=32
