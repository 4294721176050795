#!/bin/bash
# round 5, call 22: whole GPU suite on the clean build (sector certificate inside the grid searches, complex64-only magnitude bound),
# then the config-4 rates
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -6
for dt in f64 f32; do for dm in mindist slicer; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype $dt --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $dt $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done; done
python scripts/experiments/r05_psk_rates.py
