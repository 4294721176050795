#!/bin/bash
# round 4, call 25: packed complex arithmetic (v_pk_*_f32) in the complex64 planar kernels: parity, rates, counters
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_planar_f32.py tests/test_gpu_f64_kernel.py -m gpu -q --timeout=600 -x 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert|Error" | tail -15
for dm in mindist slicer; do
for t in 0 257 259 260 512 256; do
  python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod $dm --dtype f32 --config c4 --batch 262144 --opt f32_planar=1 --opt f64_threads=$t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f32 $dm planar threads=$t', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
done
done
python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --demod mindist --dtype f64 --config c4 --batch 262144 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 f64 mindist', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms_per_launch'], 'ser %.6f' % d['ser'])"
timeout 600 python scripts/bench_f64_family.py --dtype f32 > gpurun_out/f32_family_rates.json 2> gpurun_out/f32_family.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/f32_family_rates.json"))
for k, r in d.items():
    print(k, " ".join("%s %.4g" % (n, v["realizations_per_s"]) for n, v in r.items() if isinstance(v, dict)), "x%.2f" % r.get("fast_over_generic", 0))
PY
TAG=c4p260 bash scripts/prof_one.sh c4 --dtype f32 --demod mindist --batch 262144 --opt f32_planar=1 --opt f64_threads=260 > /dev/null 2>&1
echo "== c4p260"; for d in gpurun_out/prof_c4p260_*; do python scripts/pmc_summary.py $d k_run_mimo_ofdm_planar 262144 2>/dev/null | grep -v "^None\|^('void"; done
true
