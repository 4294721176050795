#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_legacy.py -m gpu -q --timeout=900 -x -k "tdl or fuzz" 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c3 --batch 131072 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 f64', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done
