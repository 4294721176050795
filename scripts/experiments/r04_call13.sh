#!/bin/bash
# round 4, call 13: k_bd_link specialised by user count and demodulator path (no spills) -- BD tests, f6 rates; whole suite;
# family rates with the final kernels
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -5
for dt in f32 f64; do
python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --dtype $dt --config f6 --batch 524288 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f6 $dt', '%.4g /s' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
timeout 600 python scripts/bench_f64_family.py > gpurun_out/f64_family_rates.json 2> gpurun_out/f64_family_rates.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/f64_family_rates.json'))
for k,v in d.items():
    print(k, ' '.join('%s %.4g' % (n, v[n]['realizations_per_s']) for n in ('fast_mindist','fast_slicer','generic_mindist') if n in v), 'x%.2f' % v.get('fast_over_generic', 0))
PY
