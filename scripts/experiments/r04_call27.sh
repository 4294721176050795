#!/bin/bash
# round 4, call 27: profiles of config 4 under the final kernel names (complex64 planar default + the matrix-core kernel behind
# f32_mfma, complex128), the family rates in both arithmetics
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/prof_r04.sh c4 c4md c4_mfma c4md_mfma c4_f64 c4_f64sl > gpurun_out/prof_r04.log 2>&1; tail -6 gpurun_out/prof_r04.log
timeout 600 python scripts/bench_f64_family.py --dtype f32 > gpurun_out/f32_family_rates.json 2> gpurun_out/f32_family.err
timeout 600 python scripts/bench_f64_family.py > gpurun_out/f64_family_rates.json 2> gpurun_out/f64_family.err
python - <<'PY'
import json
for f in ("f32", "f64"):
    d = json.load(open("gpurun_out/%s_family_rates.json" % f))
    for k, r in d.items():
        print(f, k, " ".join("%s %.4g" % (n, v["realizations_per_s"]) for n, v in r.items() if isinstance(v, dict)), "x%.2f" % r.get("fast_over_generic", 0))
PY
python scripts/collect_profiles.py r04 2>&1 | tail -12
