#!/bin/bash
# round 4, call 3: the complex128 kernel family -- parity tests, then rates fast vs generic per geometry
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_f64_kernel.py tests/test_demod_cert.py tests/test_gpu_bench_contract.py -m gpu -q --timeout=900 2>&1 | tail -15
timeout 600 python scripts/bench_f64_family.py > gpurun_out/f64_family_rates.json 2> gpurun_out/f64_family_rates.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/f64_family_rates.json'))
for k,v in d.items():
    print(k, ' '.join('%s %.4g' % (n, v[n]['realizations_per_s']) for n in ('fast_mindist','fast_slicer','generic_mindist') if n in v), 'x%.2f' % v.get('fast_over_generic', 0))
PY
python bench.py --steps 6 --warmup 2 --no-cpu --pmc off --single-demod --dtype f64 --config c4 --batch 262144 2>/dev/null | tail -c 600
