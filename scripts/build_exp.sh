#!/bin/bash
# The -DMCLE_EXPERIMENTS variant of the library (section ablations / timing bounds: WRONG results by construction, never the product
# build) -> scripts/experiments/bin/libmcle_exp.so, built here (hipcc cross-compiles) and shipped to the GPU box with the snapshot;
# select it with MCLE_LIBRARY=$PWD/scripts/experiments/bin/libmcle_exp.so.  Only the translation units that look at the macro are
# recompiled; every other object is the product build's.  EXP_DEFS="-DNAME=VALUE ..." EXP_NAME=libmcle_<tag>.so: a second variant
# with extra macros (A/B of compile-time settings).
set -e
cd "$(dirname "$0")/../pyphysim_amd/csrc"
make -j8 > /dev/null
OUT=../../scripts/experiments/bin; mkdir -p $OUT /tmp/mcle_exp
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast -DMCLE_EXPERIMENTS $EXP_DEFS"
EXP="capi pipeline_mimo_qw pipeline_mimo_pw pipeline_mimo_fw pipeline_mimo_planar kernels_ia kernels_bd"
for f in $EXP; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/mcle_exp/$f.o & done; wait
OBJS=""
for o in *.o; do b=${o%.o}; if echo " $EXP " | grep -q " $b "; then OBJS="$OBJS /tmp/mcle_exp/$o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/${EXP_NAME:-libmcle_exp.so} $OBJS -ldl
ls -la $OUT/${EXP_NAME:-libmcle_exp.so}
