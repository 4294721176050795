#!/bin/bash
# Round-5 profiles of every fused pipeline in both arithmetics (runs on the GPU box): per tag a `rocprofv3 --kernel-trace --stats`
# run and four separate `--pmc` passes (counters are never combined with tracing) of `bench.py --single-demod` on that
# (config, dtype, demodulator); scripts/collect_profiles.py r05 condenses gpurun_out/prof_<tag>_* into profiles/r05/.
# Every step is ONE dispatch of the dominant kernel at these batches (round 5: the record buffers of the TDL pipelines hold 2 / 4 GiB,
# complex128 config 4 slices at 2^18); scripts/collect_profiles.py sums the counters over every dispatch of the run anyway and
# divides by the realizations they cover.
# usage: bash scripts/prof_r05.sh [tag ...]        (default: all)
mkdir -p gpurun_out; export TMPDIR=/tmp
declare -A SPEC=(
  [c4_f64]="--config c4 --dtype f64 --demod mindist --batch 262144"
  [c4_f64sl]="--config c4 --dtype f64 --demod slicer --batch 262144"
  [c4]="--config c4 --dtype f32 --demod slicer --batch 262144"
  [c4md]="--config c4 --dtype f32 --demod mindist --batch 262144"
  [c4_mfma]="--config c4 --dtype f32 --demod slicer --batch 262144 --opt f32_mfma=1"
  [c4md_mfma]="--config c4 --dtype f32 --demod mindist --batch 262144 --opt f32_mfma=1"
  [c3]="--config c3 --dtype f32 --batch 262144"
  [c3_f64]="--config c3 --dtype f64 --batch 131072"
  [c3_mfma]="--config c3 --dtype f32 --batch 262144 --opt tdl_kernel=1"
  [c3_f64_batch]="--config c3 --dtype f64 --batch 131072 --opt tdl_kernel=1"
  [c2]="--config c2 --dtype f32 --batch 65536"
  [c2_f64]="--config c2 --dtype f64 --batch 16384"
  [f1]="--config f1 --dtype f32 --demod slicer --batch 98304"
  [f1_coop]="--config f1 --dtype f32 --demod slicer --batch 98304 --opt mimo_tdl_kernel=1"
  [f1_f64_coop]="--config f1 --dtype f64 --batch 32768 --opt mimo_tdl_kernel=1"
  [c5]="--config c5 --dtype f32 --demod slicer --batch 1048576"
  [f6]="--config f6 --dtype f32 --batch 524288"
  [f1_f64]="--config f1 --dtype f64 --batch 98304"
  [c5_f64]="--config c5 --dtype f64 --batch 262144"
  [f6_f64]="--config f6 --dtype f64 --batch 131072"
)
tags=${@:-c4_f64 c4_f64sl c4 c4md c4_mfma c4md_mfma c3 c3_f64 c3_mfma c3_f64_batch c2 c2_f64 f1 f1_coop c5 f6 f1_f64 f1_f64_coop c5_f64 f6_f64}
for tag in $tags; do
  spec=${SPEC[$tag]}
  echo "{\"tag\": \"$tag\", \"bench_args\": \"$spec\"}" > gpurun_out/prof_${tag}_meta.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -o $tag -- python bench.py $spec --steps 10 --warmup 2 --no-cpu --pmc off --single-demod > gpurun_out/prof_${tag}_stats.log 2>&1
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY"; do
    t=$(echo $pmc | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/prof_${tag}_$t -o $tag -- python bench.py $spec --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/prof_${tag}_$t.log 2>&1
  done
  echo "profiled $tag"
done
