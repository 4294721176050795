#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bd.py -m gpu -q --timeout=600 -x 2>&1 | tail -3
for opt in "bd_runtime_solve=0" "bd_runtime_solve=1"; do
for dt in f32 f64; do
  python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype $dt --config f6 --single-demod --batch 131072 --opt $opt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$opt $dt', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g/prof_f6 -o f6 -- python bench.py --config f6 --dtype f32 --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --batch 131072 > /dev/null 2>&1
head -4 gpurun_out/r03g/prof_f6/*/f6_kernel_stats.csv 2>/dev/null | cut -c1-60,200-330 ; head -4 gpurun_out/r03g/prof_f6/f6_kernel_stats.csv 2>/dev/null | cut -c1-60,200-330
