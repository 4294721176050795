#!/bin/bash
# round 4, call 32: counters of the one-realization-per-wavefront config-3 kernel next to the matrix-core kernel's
export TMPDIR=/tmp
TAG=c3w bash scripts/prof_one.sh c3 --dtype f32 --batch 262144 --opt tdl_wave=1 > /dev/null 2>&1
TAG=c3w64 bash scripts/prof_one.sh c3 --dtype f64 --batch 131072 --opt tdl_wave=1 > /dev/null 2>&1
for t in c3w c3w64; do echo "== $t"; n=262144; [ $t = c3w64 ] && n=131072; for d in gpurun_out/prof_${t}_*; do python scripts/pmc_summary.py $d k_run_ofdm_tdl_wave $n 2>/dev/null | grep -v "^None\|^('void"; done; grep -h "k_run_ofdm_tdl_wave\|k_tdl_symbol" gpurun_out/prof_${t}_stats/*kernel_stats.csv | cut -c1-200; done
true
