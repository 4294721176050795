#!/bin/bash
# usage (on the GPU box): TAG=<name> bash scripts/prof_one.sh <cfg> [extra bench args]   -> gpurun_out/prof_<TAG>_*/
cfg=${1:-c4}; shift
tag0=${TAG:-$cfg}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag0}_stats -o $tag0 -- python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --pmc off --single-demod "$@" > gpurun_out/prof_${tag0}_stats.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/prof_${tag0}_$tag -o $tag0 -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 "$@" > gpurun_out/prof_${tag0}_$tag.log 2>&1
done
python scripts/collect_profiles.py scratch 2>/dev/null | head -40
