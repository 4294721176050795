mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
for cfg in "c4 --demod slicer" "c4 --demod mindist" "c3" "c2"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu --config $cfg > gpurun_out/bench_${cfg// /_}.log 2>&1
  echo "== $cfg rc=$?"; tail -1 gpurun_out/bench_${cfg// /_}.log | cut -c1-600
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o c4 -- python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/prof_stats.log 2>&1
echo "prof stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o c4 -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_fetch.log 2>&1
echo "prof fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o c4 -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_write.log 2>&1
echo "prof write rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_sq -o c4 -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_sq.log 2>&1
echo "prof sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_lds -o c4 -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_lds.log 2>&1
echo "prof lds rc=$?"
find gpurun_out -name "*.csv" | head -30
