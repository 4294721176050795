#!/bin/bash
# round 4, call 49: grid oversubscription sweep for the other fused pipelines (bench.py's other_workloads batches and full batches)
export TMPDIR=/tmp
for cfg in "c2 f32" "c2 f64" "c5 f32" "c5 f64" "f1 f32" "f1 f64" "f6 f32" "f6 f64"; do
  set -- $cfg
  line="$1 $2:"
  for o in 0 1 2 4 8 16; do
    v=$(python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype $2 --config $1 --opt grid_oversub=$o 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g' % d['value'])")
    line="$line  o=$o $v"
  done
  echo "$line"
done
