#!/bin/bash
# usage: lib_ab.sh <lib.so under scripts/experiments/bin> ...  -- the headline and the walks' bench legs on the product library and on
# variant builds (scripts/build_exp.sh with EXP_DEFS / EXP_NAME), alternating -> gpurun_out/lib_ab.log
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/lib_ab.log
for rep in 1 2; do
for tag in c4_f64 c5_f64 f6_f64 c3_f64 c3 f1; do
  ARGS="$(python bench.py --profile-spec $tag) --no-cpu --pmc off --single-demod"
  [ $tag = c4_f64 ] && ARGS="--no-cpu --pmc off --single-demod --demod mindist"
  for lib in product "$@"; do
    if [ $lib = product ]; then unset MCLE_LIBRARY; else export MCLE_LIBRARY=$PWD/scripts/experiments/bin/$lib; fi
    line=$(timeout 300 python bench.py --steps 10 --warmup 3 $ARGS 2>/dev/null | tail -1)
    echo "$tag $lib $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4e realizations/s  %.3f ms per launch" % (d["value"], d["roofline"]["kernel_ms_per_launch"]))')" | tee -a gpurun_out/lib_ab.log
  done
done
done
