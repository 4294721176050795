#!/bin/bash
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -2
for dm in mindist slicer; do
for cfg in c2 c3 c5 f1 f6; do
b=131072; [ $cfg = c2 ] && b=16384; [ $cfg = f1 ] && b=98304; [ $cfg = c5 ] && b=262144
python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --single-demod --demod $dm --dtype f64 --config $cfg --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg f64 $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
done; done
