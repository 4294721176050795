#!/bin/bash
# round 4, call 35: counters of the wave kernel (complex64) after call 33's changes
export TMPDIR=/tmp
TAG=c3w bash scripts/prof_one.sh c3 --dtype f32 --batch 262144 --opt tdl_wave=1 > /dev/null 2>&1
for d in gpurun_out/prof_c3w_SQ_WAVE_CYCLES gpurun_out/prof_c3w_SQ_LDS_BANK_CONFLICT; do python scripts/pmc_summary.py $d k_run_ofdm_tdl_wave 262144 2>/dev/null | grep -v "^None"; done
grep -h "k_run_ofdm_tdl_wave\|k_tdl_symbol" gpurun_out/prof_c3w_stats/*kernel_stats.csv | cut -c1-60,200-300
