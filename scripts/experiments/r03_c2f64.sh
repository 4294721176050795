#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x -k "flat or jakes or tdl or legacy or operators or fuzz or slices" 2>&1 | tail -2
for cfg in c2 c3 f1; do
python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --single-demod --dtype f64 --config $cfg --batch $([ $cfg = c2 ] && echo 16384 || echo 131072) 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg f64', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done
python scripts/bench_operators.py --dtype=f64 --only=jakes 2>/dev/null | tail -3
