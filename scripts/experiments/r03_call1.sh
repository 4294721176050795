#!/bin/bash
# round 3, first GPU call: f64 datapath microbenchmarks + complex128 baselines of every fused pipeline
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
hipcc -O2 --offload-arch=gfx950 scripts/experiments/f64_rates.hip -o /tmp/f64_rates 2>/dev/null && timeout 300 /tmp/f64_rates > gpurun_out/r03a/f64_rates.txt 2>&1
echo "f64_rates rc=$?"
for cfg in c4 c3 c2 c5 f1 f6; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --dtype f64 --config $cfg --single-demod > gpurun_out/r03a/bench_f64_$cfg.json 2> gpurun_out/r03a/bench_f64_$cfg.err
  echo "== f64 $cfg rc=$?"; python - gpurun_out/r03a/bench_f64_$cfg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %.4g realizations/s  kernel ms %.3f per %d" % (d["value"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["realizations_per_launch"]))
except Exception as e:
    print("   (no json)", e)
PY
done
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --dtype f64 --config c4 --single-demod --demod mindist > gpurun_out/r03a/bench_f64_c4md.json 2>/dev/null
tail -c 300 gpurun_out/r03a/bench_f64_c4md.json | head -c 300; echo
# counters of the f64 config-4 kernel (k_run_mimo_ofdm<double,...>)
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/r03a/pmc_c4f64_$tag -o c4 -- python bench.py --config c4 --dtype f64 --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/r03a/pmc_c4f64_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
cat gpurun_out/r03a/f64_rates.txt | head -60
