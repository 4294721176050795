#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_mfma.py tests/test_gpu_fuzz.py tests/test_gpu_pipelines.py -m gpu -q --timeout=900 -x -k "tdl or fuzz" 2>&1 | tail -2
run() { python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist --dtype f32 --config c3 --batch 131072 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"; }
run
run --opt tdl_mfma_waves=3
run
run --batch 2097152
for pmc in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/r03h/pmc_$tag -o c3 -- python bench.py --config c3 --dtype f32 --steps 3 --warmup 1 --no-cpu --pmc off --single-demod --preroll-ms 0 --batch 131072 > /dev/null 2>&1
done
python scripts/pmc_summary.py gpurun_out/r03h k_run_ofdm_tdl_mfma 131072
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h/prof_c3 -o c3 -- python bench.py --config c3 --dtype f32 --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --batch 131072 > /dev/null 2>&1
head -4 gpurun_out/r03h/prof_c3/*/c3_kernel_stats.csv 2>/dev/null | cut -c1-60,230-330 ; head -4 gpurun_out/r03h/prof_c3/c3_kernel_stats.csv 2>/dev/null | cut -c1-60,230-330
