#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test-suite, then the bench for every config.
# usage: bash scripts/gpu_check.sh [quick|full|prof|profonly] [pytest -k expression]
mode=${1:-quick}
kexpr=${2:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$mode" != "profonly" ]; then
  if [ -n "$kexpr" ]; then
    timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -k "$kexpr" > gpurun_out/pytest_gpu.log 2>&1
  else
    timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1
  fi
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
  for cfg in "c4 --demod slicer" "c3" "c2" "c5" "f1" "f6"; do
    name=${cfg// /_}; name=${name//--demod_/}
    timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --config $cfg > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
    echo "== $cfg rc=$?"; python - "gpurun_out/bench_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %.4g realizations/s  ms/step %.3f  kernel ms %.3f  fp32 frac %.3f  ser %.5f  %s" % (
        d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"], d["ser"],
        d["config"].get("demod_rates")))
except Exception as e:
    print("   (no json)", e)
PY
  done
fi
if [ "$mode" = "full" ] || [ "$mode" = "prof" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
  echo "default bench rc=$?"; tail -c 1200 gpurun_out/bench_default.json
  # the driver's multi-GPU launch line with one rank: RCCL initialised, the counter all-reduce inside the timed region
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_torchrun_1rank.log 2>&1
  echo "torchrun 1-rank rc=$?"; tail -c 600 gpurun_out/bench_torchrun_1rank.log
fi
if [ "$mode" = "prof" ] || [ "$mode" = "profonly" ]; then
  for cfg in ${PROF_CFGS:-c4 c3 c2 f1 c5 f6}; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${cfg}_stats -o $cfg -- python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu --pmc off --single-demod > gpurun_out/prof_${cfg}_stats.log 2>&1
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY"; do
      tag=$(echo $pmc | cut -d' ' -f1)
      timeout 600 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/prof_${cfg}_$tag -o $cfg -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/prof_${cfg}_$tag.log 2>&1
      echo "pmc $cfg $tag rc=$?"
    done
  done
  if [ -n "$PROF_SKIP_EXTRAS" ]; then exit 0; fi
  timeout 300 python scripts/bench_ia_solvers.py > gpurun_out/ia_solvers.json 2> gpurun_out/ia_solvers.err
  timeout 300 python scripts/bench_mimo_schemes.py > gpurun_out/mimo_schemes.json 2> gpurun_out/mimo_schemes.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_operators_stats -o operators -- python scripts/bench_operators.py > gpurun_out/operators.json 2> gpurun_out/operators.err
  timeout 300 python scripts/bench_simulators.py > gpurun_out/simulators.json 2> gpurun_out/simulators.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f1staged_stats -o f1staged -- python scripts/bench_staged_f1.py --batch 2048 --steps 5 > gpurun_out/staged_f1.json 2> gpurun_out/prof_f1staged.log
fi
