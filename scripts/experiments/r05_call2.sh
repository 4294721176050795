#!/bin/bash
# round 5, call 2: counters of the wavefront-per-antenna f1 kernel (complex64, then complex128)
export TMPDIR=/tmp
for dt in f32 f64; do
  TAG=f1w_$dt bash scripts/prof_one.sh f1 --dtype $dt --batch 98304 > /dev/null 2>&1
  for d in gpurun_out/prof_f1w_${dt}_*; do :; done
  python scripts/pmc_summary.py gpurun_out "k_run_mimo_ofdm_tdl_wave" 98304 2>&1 | tail -32
  rm -rf gpurun_out/prof_f1w_${dt}_*/*/*.db 2>/dev/null
  mkdir -p gpurun_out/keep_$dt; mv gpurun_out/prof_f1w_${dt}_* gpurun_out/keep_$dt/ 2>/dev/null
done
