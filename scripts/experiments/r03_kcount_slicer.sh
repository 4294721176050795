export TMPDIR=/tmp
python scripts/bench_staged_c4.py --seconds 1 --demod slicer | python -c "import json,sys; d=json.load(sys.stdin); print('staged f32 slicer', d['realizations_per_s'], d['frac'], d['ms_per_pass'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_staged_c4_sl -o staged_c4_sl -- python scripts/bench_staged_c4.py --seconds 0.3 --demod slicer > /dev/null 2>&1
grep k_count gpurun_out/prof_staged_c4_sl/staged_c4_sl_kernel_stats.csv | cut -c1-40,200-
