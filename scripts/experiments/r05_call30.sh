#!/bin/bash
# round 5, call 30: the section table of the headline kernel again on the chained-FMA build (experiments library), and the H(f) A/B
# with the product's class-position form as the VALU side
export MCLE_LIBRARY=$PWD/pyphysim_amd/csrc/libmcle_exp.so
bash scripts/experiments/r05_c4_sections.sh
unset MCLE_LIBRARY
scripts/experiments/bin/hf_mfma_ab
