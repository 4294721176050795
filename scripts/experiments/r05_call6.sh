#!/bin/bash
# round 5, call 6: counters of the complex64 wavefront f1 kernel after the packed-complex rewrite (one launch per step: 83 886 realizations)
export TMPDIR=/tmp
TAG=f1w_f32 bash scripts/prof_one.sh f1 --dtype f32 --batch 83886 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out "k_run_mimo_ofdm_tdl_wave" 83886 2>&1 | tail -28
rm -rf gpurun_out/prof_f1w_f32_*
