#!/bin/bash
# round 4, call 36: is the wave-asynchronous config-3 kernel instruction-fetch bound?  (I-cache counters, against the planar config-4 kernel)
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | cut -c1-160 | sort -u | head -40
cd $GRAFT_REPO_ROOT
for spec in "c3w --config c3 --dtype f32 --batch 262144 --opt tdl_wave=1:k_run_ofdm_tdl_wave" "c3b --config c3 --dtype f32 --batch 262144 --opt tdl_wave=0:k_run_ofdm_tdl_mfma" "c4p --config c4 --dtype f32 --demod mindist --batch 262144:k_run_mimo_ofdm_planar" "c3w64 --config c3 --dtype f64 --batch 131072 --opt tdl_wave=1:k_run_ofdm_tdl_wave"; do
  tag=$(echo $spec | cut -d' ' -f1); needle=${spec##*:}; args=${spec%:*}; args=${args#* }
  for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_SALU"; do
    t=$(echo $pmc | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/ic_${tag}_$t -o $tag -- python bench.py $args --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/ic_${tag}_$t.log 2>&1
  done
  echo "== $tag"; for d in gpurun_out/ic_${tag}_*/; do python scripts/pmc_summary.py $d $needle 262144 2>/dev/null | grep -v "^None\|^('void\|busy_chip\|kernel cycles"; done
done
