#!/bin/bash
# round 6: the packed walk against the grid (option grid_oversub = workgroups per resident slot) -> gpurun_out/walk_grid_sweep.log (usage: walk_grid_sweep.sh [profile tag ...])
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/walk_grid_sweep.log
for tag in ${@:-c5 c5_f64 f6 f6_f64}; do
  ARGS="$(python bench.py --profile-spec $tag) --no-cpu --pmc off --single-demod"
  for g in 0 1 2 4 8; do
    line=$(timeout 300 python bench.py --steps 10 --warmup 3 $ARGS --opt grid_oversub=$g 2>/dev/null | tail -1)
    echo "$tag grid_oversub=$g $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4e realizations/s  %.3f ms per launch" % (d["value"], d["roofline"]["kernel_ms_per_launch"]))')" | tee -a gpurun_out/walk_grid_sweep.log
  done
done
