#!/bin/bash
# round 6: the packed walk (csrc/walk_f64.hpp) against the per-realization walks of rounds 2-5 (option walk_legacy=1) at the bench
# legs' batches, both arithmetics and demodulators -> gpurun_out/walk_ab.log
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/walk_ab.log
for dt in f64 f32; do
for cfg in c5 f6; do
  for demod in slicer mindist; do
    tag=$cfg; [ $dt = f64 ] && tag=${cfg}_f64
    ARGS="$(python bench.py --profile-spec $tag | sed "s/--demod slicer/--demod $demod/") --no-cpu --pmc off --single-demod"
    for legacy in 1 0; do
      line=$(timeout 300 python bench.py --steps 10 --warmup 3 $ARGS --opt walk_legacy=$legacy 2>/dev/null | tail -1)
      echo "$cfg $dt $demod walk_legacy=$legacy $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4e realizations/s  %.3f ms per launch" % (d["value"], d["roofline"]["kernel_ms_per_launch"]))')" | tee -a gpurun_out/walk_ab.log
    done
  done
done
done
