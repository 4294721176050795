#!/bin/bash
# round 5, call 18: config 3's wavefront kernel in complex128 at 2048 points (two wavefronts per workgroup): parity, family rates
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_tdl_wave.py tests/test_gpu_pipelines.py -m gpu -q --timeout=900 -k "tdl or ofdm" 2>&1 | tail -4
timeout 600 python scripts/bench_tdl_family.py > gpurun_out/tdl_family_rates.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/tdl_family_rates.json"))
for k,v in d.items():
    print(k, "wave %.4g batched %.4g x%.2f" % (v["wave"]["realizations_per_s"], v["batched"]["realizations_per_s"], v["wave_over_batched"]), "ser", round(v["wave"]["ser"],6), round(v["batched"]["ser"],6))
PY
