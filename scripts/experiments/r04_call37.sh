#!/bin/bash
# round 4, call 37: wave kernel, complex64 at a three- / four-wavefront register bound; counters
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tdl_wave.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -4
for w in 0 1 4; do
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --dtype f32 --config c3 --batch 262144 --opt tdl_wave=$w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 f32 tdl_wave=$w', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
TAG=c3w bash scripts/prof_one.sh c3 --dtype f32 --batch 262144 --opt tdl_wave=1 > /dev/null 2>&1
for d in gpurun_out/prof_c3w_SQ_WAVE_CYCLES gpurun_out/prof_c3w_SQ_LDS_BANK_CONFLICT; do python scripts/pmc_summary.py $d k_run_ofdm_tdl_wave 262144 2>/dev/null | grep -v "^None\|^('void"; done
