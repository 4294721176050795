#!/bin/bash
mkdir -p gpurun_out/r03d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=600 -x > gpurun_out/r03d/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03d/pytest_gpu.log
for dm in slicer mindist; do
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype f64 --config c4 --single-demod --demod $dm > gpurun_out/r03d/bench_f64_c4_${dm}.json 2> gpurun_out/r03d/err.txt
  echo "== f64 c4 $dm rc=$?"; python - gpurun_out/r03d/bench_f64_c4_${dm}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %.4g realizations/s  kernel ms %.3f per %d  ser %.6f" % (d["value"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["realizations_per_launch"], d["ser"]))
except Exception as e:
    print("   (no json)", e)
PY
done
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/r03d/pmc_$tag -o c4 -- python bench.py --config c4 --dtype f64 --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/r03d/pmc_$tag.log 2>&1
done
python scripts/pmc_summary.py gpurun_out/r03d k_run_mimo_ofdm_f64 65536
