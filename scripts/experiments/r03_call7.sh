#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_f64_kernel.py tests/test_gpu_operators.py -m gpu -q --timeout=600 -x 2>&1 | tail -2
for th in 512 256; do
for dm in slicer mindist; do
  python bench.py --steps 10 --warmup 2 --no-cpu --pmc off --dtype f64 --config c4 --single-demod --demod $dm --batch 262144 --opt f64_threads=$th 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $th $dm', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], d['ser'])"
done; done
hipcc -O3 --offload-arch=gfx950 scripts/experiments/dft16_f64_ab.hip -o /tmp/dft16_ab 2>/dev/null && /tmp/dft16_ab | tee gpurun_out/dft16_f64_ab.txt
