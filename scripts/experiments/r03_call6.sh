#!/bin/bash
mkdir -p gpurun_out/r03f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_staged_c4.py tests/test_gpu_f64_kernel.py -m gpu -q --timeout=600 -x > gpurun_out/r03f/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r03f/pytest_gpu.log
for b in 2048 8192 16384; do
python scripts/bench_staged_c4.py --batch $b --seconds 1 > gpurun_out/r03f/staged_c4_$b.json 2> gpurun_out/r03f/staged_err.txt; echo "staged rc=$?"
python -c "import json; d=json.load(open('gpurun_out/r03f/staged_c4_$b.json')); print($b, '%.4g /s' % d['realizations_per_s'], 'frac %.3f' % d['frac'], 'ms/pass %.3f' % d['ms_per_pass'], d['ser'])"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03f/prof_staged -o staged -- python scripts/bench_staged_c4.py --batch 8192 --seconds 0.3 > gpurun_out/r03f/prof_staged.log 2>&1
head -20 gpurun_out/r03f/prof_staged/*/staged_kernel_stats.csv 2>/dev/null | cut -c1-150
start=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03f/bench_default.json 2> gpurun_out/r03f/bench_default.err
echo "default bench rc=$? took $(( $(date +%s) - start )) s"; tail -c 600 gpurun_out/r03f/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03f/bench_default.json').read().strip().splitlines()[-1])
for k in ("value", "dtype", "value_is", "ms_per_step", "timed_region_s", "dtype_rates", "rates", "ser", "ser_abs_err_vs_oracle"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "achieved", "peak", "kernel", "valu_busy_chip", "traffic", "counters_source")})
print("roofline_f32", {k: d.get("roofline_f32", {}).get(k) for k in ("frac", "achieved", "kernel", "valu_busy_chip", "mfma_busy_chip")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_all_cores", {}).get("value"))
print("others", json.dumps(d.get("other_workloads"))[:1500])
PY
