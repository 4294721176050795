#!/bin/bash
# round 4, call 33: quadrant certificate (QPSK), wave kernel with register hand-over into the forward transform, compile-time
# polynomial orders, four workgroups per CU in complex64: parity, then config-3 rates (batched / wave)
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tdl_wave.py tests/test_demod_cert.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert|Error" | tail -15
for dt in f32 f64; do for w in 0 1; do
  b=262144; [ $dt = f64 ] && b=131072
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --dtype $dt --config c3 --batch $b --opt tdl_wave=$w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $dt tdl_wave=$w', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --dtype f32 --config c3 --batch 262144 --opt tdl_wave=1 --opt demod_nocert=1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 f32 wave nocert', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_pipelines.py tests/test_gpu_mfma.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
