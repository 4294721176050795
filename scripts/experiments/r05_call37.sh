#!/bin/bash
# round 5, call 37: the build with the per-geometry packed multiply-adds (planar complex64) and the chained FMAs in the flat-MIMO
# walk: the complex64 family again, the flat-MIMO schemes in both arithmetics, then the whole GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python scripts/bench_f64_family.py --dtype f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 family', {k: '%.4g' % v['fast_mindist']['realizations_per_s'] for k, v in d.items() if isinstance(v, dict) and 'fast_mindist' in v})"
timeout 300 python scripts/bench_mimo_schemes.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('flat MIMO f32', ' '.join('%s %.3f ms' % (k, v['kernel_ms']) for k,v in d.items()))"
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
