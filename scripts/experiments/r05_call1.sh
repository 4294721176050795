#!/bin/bash
# round 5, call 1: the wavefront-per-antenna f1 kernel (csrc/mimo_tdl_wave.hpp): first parity run + A/B against the cooperative kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mimo_tdl_wave.py -m gpu -q --timeout=600 -x -k "against_the_oracle and not 240 and not 120" 2>&1 | tail -15
for dt in f32 f64; do for k in 1 0 2; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $dt --config f1 --batch 98304 --opt mimo_tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f1 $dt kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
