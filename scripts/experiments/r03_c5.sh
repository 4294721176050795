#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipelines.py -m gpu -q --timeout=600 -x -k "ia" 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype f32 --config c5 --single-demod --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 f32', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g/prof_c5 -o c5 -- python bench.py --config c5 --dtype f32 --steps 10 --warmup 2 --no-cpu --pmc off --single-demod --batch 262144 > /dev/null 2>&1
head -4 gpurun_out/r03g/prof_c5/*/c5_kernel_stats.csv 2>/dev/null | cut -c1-50,150-330 ; head -4 gpurun_out/r03g/prof_c5/c5_kernel_stats.csv 2>/dev/null | cut -c1-50,150-330
