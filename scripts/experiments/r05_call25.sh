#!/bin/bash
# round 5, call 25: (1) the H(f) stage of f1 in complex64 on the matrix cores against the packed VALU form (standalone A/B,
# scripts/experiments/hf_mfma_ab.hip); (2) the build with the call-24 decisions (multiply-add H(f) at one bin per lane, k_bd_link<float>
# at three wavefronts per SIMD, 2x2 at 256 points on the generic kernel): the suites that touch them, then the rates
export TMPDIR=/tmp
scripts/experiments/bin/hf_mfma_ab
timeout 1500 python -m pytest tests/test_gpu_mimo_tdl_wave.py tests/test_gpu_bd.py tests/test_gpu_f64_kernel.py tests/test_gpu_planar_f32.py tests/test_gpu_oracle_depth.py tests/test_gpu_fuzz.py -q --timeout=900 2>&1 | tail -4
python scripts/experiments/r05_f1_shapes.py new | grep -E "fft (256|512)"
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --config f6 --dtype f32 --batch 524288 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f6 f32', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
timeout 400 python scripts/bench_f64_family.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f64 256x2x2', d['256x2x2'])"
timeout 400 python scripts/bench_f64_family.py --dtype f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f32 256x2x2', d['256x2x2'])"
