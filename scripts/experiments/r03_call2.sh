#!/bin/bash
mkdir -p gpurun_out/r03b
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/r03b/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03b/pytest_gpu.log
for cfg in c4 c3 c2 c5 f1 f6; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu --pmc off --dtype f64 --config $cfg --single-demod > gpurun_out/r03b/bench_f64_$cfg.json 2> gpurun_out/r03b/bench_f64_$cfg.err
  echo "== f64 $cfg rc=$?"; python - gpurun_out/r03b/bench_f64_$cfg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   %.4g realizations/s  kernel ms %.3f per %d  ser %.6f" % (d["value"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["realizations_per_launch"], d["ser"]))
except Exception as e:
    print("   (no json)", e)
PY
done
