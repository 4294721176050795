#!/bin/bash
# round 6: the packed complex128 walk with its registers bounded for 3 (product) and 4 wavefronts per SIMD
# (scripts/build_exp.sh with EXP_DEFS=-DMCLE_WALK_F64_WAVES=4 -> libmcle_w4.so) -> gpurun_out/walk_waves_ab.log
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/walk_waves_ab.log
for cfg in c5 f6; do
  for demod in slicer mindist; do
    ARGS="$(python bench.py --profile-spec ${cfg}_f64 | sed "s/--demod slicer/--demod $demod/") --no-cpu --pmc off --single-demod"
    for lib in libmcle_exp.so libmcle_w4.so; do
      line=$(MCLE_LIBRARY=$PWD/scripts/experiments/bin/$lib timeout 300 python bench.py --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1)
      echo "$cfg f64 $demod $lib $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4e realizations/s  %.3f ms per launch" % (d["value"], d["roofline"]["kernel_ms_per_launch"]))')" | tee -a gpurun_out/walk_waves_ab.log
    done
  done
done
