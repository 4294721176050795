#!/bin/bash
mkdir -p gpurun_out/r03e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_kernel.py -m gpu -q --timeout=600 -x > gpurun_out/r03e/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r03e/pytest_gpu.log
for th in 512 256; do
for dm in slicer mindist; do
  python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype f64 --config c4 --single-demod --demod $dm --opt f64_threads=$th 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $th $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done; done
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_VMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/r03e/pmc_$tag -o c4 -- python bench.py --config c4 --dtype f64 --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 > gpurun_out/r03e/pmc_$tag.log 2>&1
done
python scripts/pmc_summary.py gpurun_out/r03e k_run_mimo_ofdm_f64 65536
