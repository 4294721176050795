#!/bin/bash
# round 4, call 38: the wavefront kernel is config 3's default: whole suite, the config-3 bench lines, profiles of the four tags
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
for dt in f32 f64; do for k in 0 1; do
  python bench.py --steps 10 --warmup 3 --no-cpu --pmc off --dtype $dt --config c3 --opt tdl_kernel=$k 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $dt tdl_kernel=$k', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'batch', d['roofline']['realizations_per_launch'], 'frac %.3f' % d['roofline']['frac'], 'ser %.6f' % d['ser'])"
done; done
bash scripts/prof_r04.sh c3 c3_f64 c3_mfma c3_f64_batch > gpurun_out/prof_r04.log 2>&1; tail -4 gpurun_out/prof_r04.log
python scripts/collect_profiles.py r04 2>&1 | grep -E "^c3" | cut -c1-420
