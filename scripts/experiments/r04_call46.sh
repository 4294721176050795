#!/bin/bash
# round 4, call 46: persistent-grid oversubscription of the final kernels (option grid_oversub; 0 = automatic)
export TMPDIR=/tmp
for cfg in "c4 f64 mindist 262144" "c4 f32 mindist 262144" "c3 f32 mindist 524288" "c3 f64 mindist 262144"; do
  set -- $cfg
  for o in 0 1 2 4 8 16; do
    python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --single-demod --demod $3 --dtype $2 --config $1 --batch $4 --opt grid_oversub=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 oversub=$o', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
  done
done
