#!/bin/bash
# round 4, call 43: the wavefront kernel at 256 / 512 / 2048: parity, then rates per size next to the batched kernels
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tdl_wave.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR|assert" | tail -12
timeout 600 python scripts/bench_tdl_family.py > gpurun_out/tdl_family_rates.json 2> gpurun_out/tdl_family.err; tail -2 gpurun_out/tdl_family.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/tdl_family_rates.json"))
for k, r in d.items():
    print(k, "wave %.4g" % r["wave"]["realizations_per_s"], "batched %.4g" % r["batched"]["realizations_per_s"], "x%.2f" % r["wave_over_batched"], "ser %.5f %.5f" % (r["wave"]["ser"], r["batched"]["ser"]))
PY
timeout 900 python -m pytest tests/test_gpu_pipelines.py tests/test_gpu_fuzz.py tests/test_gpu_mfma.py tests/test_gpu_slices.py -m gpu -q --timeout=600 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -4
