#!/bin/bash
# round 4, call 28: twiddles of a complex64 radix-16 transform fetched at its top / per pass (258); planar-family fuzz, two offsets
export TMPDIR=/tmp
for dm in mindist slicer; do
for t in 0 258 259; do
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype f32 --config c4 --batch 262144 --opt f64_threads=$t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 $dm planar threads=$t', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
done
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_planar_f32.py tests/test_gpu_mfma.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
MCLE_FUZZ_OFFSET=31 MCLE_FUZZ_TRIALS=60 timeout 900 python -m pytest tests/test_gpu_fuzz.py -k planar_family -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert" | tail -12
MCLE_FUZZ_OFFSET=77 MCLE_FUZZ_TRIALS=60 timeout 900 python -m pytest tests/test_gpu_fuzz.py -k planar_family -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert" | tail -12
