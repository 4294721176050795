#!/bin/bash
# Round-6 profiles of every fused pipeline in both arithmetics (runs on the GPU box).  The (config, dtype, demodulator, batch, options)
# of a tag come from ONE table, bench.py: profile_specs() -- every `other_workloads` leg at ITS batch and demodulator, the headline at
# the size of one dispatch -- so that a fraction recomputed from profiles/r06/ is the one bench.py prints.
# Per tag: a `rocprofv3 --kernel-trace --stats` run and six separate `--pmc` passes (counters are never combined with tracing);
# scripts/collect_profiles.py r06 condenses gpurun_out/prof_<tag>_* into profiles/r06/.
# usage: bash scripts/prof_r06.sh [tag ...]        (default: all)
mkdir -p gpurun_out; export TMPDIR=/tmp
tags=${@:-$(python bench.py --profile-spec list)}
for tag in $tags; do
  spec=$(python bench.py --profile-spec $tag) || continue
  echo "{\"tag\": \"$tag\", \"bench_args\": \"$spec\"}" > gpurun_out/prof_${tag}_meta.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_stats -o $tag -- python bench.py $spec --steps 10 --warmup 2 --no-cpu --pmc off --single-demod > gpurun_out/prof_${tag}_stats.log 2>&1
  for pmc in "FETCH_SIZE" "WRITE_SIZE" \
     "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
     "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY" \
     "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
     "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32 SQ_BUSY_CYCLES SQ_IFETCH"; do
    t=$(echo $pmc | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/prof_${tag}_$t -o $tag -- python bench.py $spec --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/prof_${tag}_$t.log 2>&1
  done
  echo "profiled $tag"
done
