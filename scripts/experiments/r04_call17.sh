#!/bin/bash
# round 4, call 17: receive filter in registers across a thread's subcarriers (256-register forms), A/B against the previous build
export TMPDIR=/tmp
one() {
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
}
cp pyphysim_amd/csrc/libmcle.so /tmp/new.so
for rep in 1 2 3; do
  cp /tmp/new.so pyphysim_amd/csrc/libmcle.so; one "G in registers"
  cp gpurun_in/libmcle_prev.so pyphysim_amd/csrc/libmcle.so; one "G from LDS   "
done
cp /tmp/new.so pyphysim_amd/csrc/libmcle.so
timeout 600 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed" | tail -3
