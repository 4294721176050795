#!/bin/bash
# round 4, call 47: config-3 wavefront kernel, grid oversubscription: automatic against explicit values, alternating
export TMPDIR=/tmp
for o in 0 8 0 8 2 0 2 3 1; do
  python bench.py --steps 8 --warmup 2 --no-cpu --pmc off --dtype f32 --config c3 --batch 524288 --opt grid_oversub=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 f32 oversub=$o', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'])"
done
