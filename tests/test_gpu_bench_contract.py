"""GPU: the driver's contract on bench.py's ONE JSON line (N = 1, no launcher): the metric of BASELINE.json on the config-4
workload, value in complex128 with the north star's demodulator, both dtypes' rates, the roofline block of the dominant
kernel, the CPU baseline of the NumPy oracle, the staged HBM leg -- on a reduced batch so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_default_line_carries_every_contract_field():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "16384",
                          "--cpu-seconds", "1", "--cpu-multicore-seconds", "0", "--pmc", "off", "--preroll-ms", "0"],
                         env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "realizations/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and d["config"]["demod"] == "mindist" and "config 4" in d["config"]["workload"]
    assert d["value"] > 1e6 and abs(d["value"] - 16384 / d["ms_per_step"] * 1e3) <= 1e-6 * d["value"]
    assert d["value"] == d["rates"]["f64"]["mindist"] == d["dtype_rates"]["f64"]
    for dt in ("f64", "f32"):
        for demod in ("mindist", "slicer"):
            assert d["rates"][dt][demod] > 1e6
    assert d["rates"]["f32"]["slicer"] > d["rates"]["f64"]["mindist"]
    assert d["ser_abs_err_vs_oracle"] == 0.0                      # complex128: the oracle's counts on the same draws
    assert d["config"]["exchange"].startswith("none") and d["rccl"]["process_group"] is False
    for key in ("roofline", "roofline_f64", "roofline_f32"):
        r = d[key]
        assert r["unit"] == "TFLOP/s" and r["bound"] == "valu" and r["peak"] > 0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and 0.0 < r["frac"] < 1.0
        assert r["kernel_ms_per_launch"] > 0 and r["realizations_per_launch"] == 16384
    assert d["roofline"]["kernel"] == "k_run_mimo_ofdm_pw" and d["roofline_f32"]["kernel"] == "k_run_mimo_ofdm_planar"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "realizations/s" and cb["sample"]
    st = d["other_workloads"]["c4_staged"]
    assert st["b_alg_bytes_per_realization"] == 412160 and st["realizations_per_s"] > 1e6 and 0.0 < st["frac"] < 1.0
    assert abs(st["frac"] - st["b_alg_bytes_per_realization"] * st["realizations_per_s"] / 8e12) <= 1e-9
    st64 = d["other_workloads"]["c4_staged_f64"]                  # the HBM clause in the reference's precision, same run
    assert st64["dtype"] == "f64" and st64["b_alg_bytes_per_realization"] == 2 * 412160 and 0.0 < st64["frac"] < 1.0
    assert st64["ser"] == d["ser"] or abs(st64["ser"] - d["ser"]) < 0.05      # same link, other index range
    assert d["hbm_copy_GBps_measured_this_run"] > 1000.0 and st["copy_GBps_measured_this_run"] == d["hbm_copy_GBps_measured_this_run"]
    # ONE achievable-HBM figure (the best of the library's own streaming kernels, measured in this run), one key, below 1
    sr = d["hbm_stream_rates_measured_this_run"]
    assert sr["achievable_GBps"] == d["hbm_copy_GBps_measured_this_run"] == max(sr[k] for k in ("copy_GBps", "read_GBps", "triad_GBps", "write_GBps"))
    assert 3000.0 < sr["achievable_GBps"] < 8000.0
    for leg in (st, st64):
        assert 0.0 < leg["frac_of_achievable_hbm"] < 1.0 and "frac_of_measured_copy_bw" not in leg
        assert "frac_of_copy_bw_measured_this_run" not in leg
    # the tap-polynomial workloads are priced on what their kernels execute (order 2 / 5 Horner), not on the literal ray sums
    c3 = d["other_workloads"]["c3"]
    assert c3["f32"]["flops_per_realization"] == 102400 + 41600 + 55296 + 6144 + 5 * 1040 * 8 + 40 * 24
    assert c3["f64"]["flops_per_realization"] == 102400 + 41600 + 55296 + 6144 + 5 * 1040 * 20 + 40 * 36
    assert d["other_workloads"]["f1"]["f32"]["kernel"] == "k_run_mimo_ofdm_tdl_wave"
    # the matrix-core clause of the north star, driver-run: config 4 on the MFMA kernel (counters need --pmc, off in this test)
    mf = d["other_workloads"]["c4_f32_mfma"]
    assert mf["kernel"] == "k_run_mimo_ofdm_mfma" and mf["realizations_per_s"] > 1e6 and abs(mf["ser"] - d["ser"]) < 0.05
    assert d["roofline"]["hbm"].get("copy_GBps", 0) > 1000.0 or d["roofline"]["hbm"]["measured_bytes_per_realization"] is None
    for cfg in ("c2", "c3", "c5", "f1", "f6"):
        for dt in ("f64", "f32"):
            assert d["other_workloads"][cfg][dt]["realizations_per_s"] > 0, (cfg, dt)
    # round 6: fractions against every HBM denominator; the SURVEY-style flop fraction next to the implementation-sized one
    for leg in (st, st64):
        fa = leg["frac_against"]
        assert set(("copy", "read", "triad", "write", "achievable", "guide_copy_6290", "spec_8000")) <= set(fa)
        assert abs(fa["achievable"] - leg["frac_of_achievable_hbm"]) <= 1e-12 and fa["copy"] >= fa["achievable"] > fa["spec_8000"]
    for dt in ("f64", "f32"):
        f1 = d["other_workloads"]["f1"][dt]
        assert 0.0 < f1["flop_frac_survey_model"] < f1["flop_frac"] < 1.0
    assert d["roofline"]["valu_busy_chip"] is None or 0.0 < d["roofline"]["valu_busy_chip"] <= 1.0
