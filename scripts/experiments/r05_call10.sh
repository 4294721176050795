#!/bin/bash
# round 5, call 10: the re-keyed bench line (achievable-HBM figure from the library's own streaming kernels, flop models of what runs,
# counter sums over every dispatch, the matrix-core leg with live MFMA-busy) -- contract test, then the default line
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q --timeout=900 2>&1 | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "default bench rc=$?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.4g frac %.4f" % (d["value"], d["roofline"]["frac"]))
print("stream", d.get("hbm_stream_rates_measured_this_run"))
ow=d["other_workloads"]
for k in ("c2","c3","c5","f1","f6"):
    print(k, {dt: ("%.3g" % ow[k][dt]["realizations_per_s"], "%.3f" % ow[k][dt]["flop_frac"]) for dt in ("f64","f32")})
print("mfma", {k: ow["c4_f32_mfma"].get(k) for k in ("realizations_per_s","mfma_busy_chip","valu_busy_chip","counters_source","error")})
for k in ("c4_staged","c4_staged_f64"):
    print(k, {q: ow[k].get(q) for q in ("realizations_per_s","frac","frac_min_rule","frac_of_achievable_hbm","measured_over_b_alg")})
r=d["roofline"]; print({k: r.get(k) for k in ("valu_busy_chip","valu_wave_insts_per_realization","dispatches_per_step","counters_source")})
PY
