#!/bin/bash
# round 4, call 23: the planar family in complex64 (k_run_mimo_ofdm_planar<float>): parity, then rates per geometry next to
# the generic kernel and -- at (1024, 4x4) -- next to the matrix-core kernel; then the whole suite
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_planar_f32.py tests/test_gpu_f64_kernel.py tests/test_gpu_pipelines.py -m gpu -q --timeout=600 -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert|Error" | tail -15
timeout 600 python scripts/bench_f64_family.py --dtype f32 > gpurun_out/f32_family_rates.json 2> gpurun_out/f32_family.err; tail -3 gpurun_out/f32_family.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/f32_family_rates.json"))
for k, r in d.items():
    print(k, " ".join("%s %.4g" % (n, v["realizations_per_s"]) for n, v in r.items() if isinstance(v, dict)), "x%.2f" % r.get("fast_over_generic", 0))
PY
for t in 0 257 512 256; do
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist --dtype f32 --config c4 --batch 262144 --opt f32_planar=1 --opt f64_threads=$t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 planar threads=$t', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -8
