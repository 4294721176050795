#!/bin/bash
# round 6: the headline workload (config 4, complex128, 2^20 realizations per step) on the quarter-wave kernel (default) and on the same
# decomposition with the decode on the matrix cores (pipeline_mimo_pw.hip, NW = 4: f64_threads=263; 264 = two wavefronts per SIMD)
# -> gpurun_out/pw_ab.log
export TMPDIR=/tmp
mkdir -p gpurun_out; : > gpurun_out/pw_ab.log
for rep in 1 2; do
for demod in mindist slicer; do
  for thr in 0 263 264; do
    line=$(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --pmc off --single-demod --demod $demod --opt f64_threads=$thr 2>/dev/null | tail -1)
    echo "c4 f64 $demod f64_threads=$thr $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4e realizations/s  %.3f ms per launch  frac %.4f  ser %.6f" % (d["value"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"], d.get("ser", -1)))')" | tee -a gpurun_out/pw_ab.log
  done
done
done
