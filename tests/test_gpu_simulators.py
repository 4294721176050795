"""GPU: the ready-made simulators (BatchedSimulationRunner + fused pipelines) end to end through the
kept SimulationRunner / SimulationResults surface."""
import math

import numpy as np
import pytest

from helpers import relerr
from oracle import chains
from pyphysim_amd import simulators
from pyphysim_amd.modulators import QAM
from pyphysim_amd.simulations import SimulationResults

pytestmark = pytest.mark.gpu


def test_awgn_simulator_matches_theory_and_oracle(engine, tmp_path):
    sim = simulators.AwgnSimulator(SNR=[0.0, 6.0, 12.0, 18.0], modulator="qam", M=16, NSymbs=10000, rep_max=200,
                                   seed=5, batch_size=64, engine=engine, common_random_numbers=True)
    sim.set_results_filename(str(tmp_path / "awgn_{modulator}"))
    sim.simulate()
    ser = np.array(sim.results.get_result_values_list("ser"))
    theory = QAM(16, engine=engine).calcTheoreticalSER(np.array([0.0, 6.0, 12.0, 18.0]))
    # reference tests/modulators_package_test.py:299-305 style check (2e6 symbols per point)
    assert np.all(np.abs(ser - theory) < 0.02 * theory + 2e-4)
    assert sim.runned_reps == [200] * 4 and sim.results["ser"][0].num_updates == 200
    assert set(sim.results.get_result_names()) == {"symbol_errors", "num_symbols", "bit_errors", "num_bits", "ber",
                                                   "ser", "elapsed_time", "num_skipped_reps"}
    assert sim.results.get_result_values_list("num_symbols") == [200 * 10000] * 4
    lo, hi = sim.results.get_result_values_confidence_intervals("ser", P=95)[1]
    assert lo < ser[1] < hi
    # common random numbers against the oracle on the first realizations (f32: tiny boundary effects)
    want = sum(chains.chain_awgn(chains.PhiloxRng(5, r), "qam", 16, 10000, 6.0)["symbol_errors"] for r in range(8))
    got = engine.run_awgn(10000, 10 ** -0.6, 5, 0, 8, dtype="f32", method=1)["sym_errors"]
    assert abs(got - want) <= 2
    # saved results load back; partial results allow a resume with a larger rep_max
    back = SimulationResults.load_from_file(str(tmp_path / "awgn_16-QAM.pickle"))
    assert np.allclose(back.get_result_values_list("ser"), ser)
    again = simulators.AwgnSimulator(SNR=[0.0, 6.0, 12.0, 18.0], modulator="qam", M=16, NSymbs=10000, rep_max=264,
                                     seed=5, batch_size=64, engine=engine, common_random_numbers=True)
    again.set_results_filename(str(tmp_path / "awgn_{modulator}"))
    again.simulate()
    assert again.runned_reps == [264] * 4
    fresh = simulators.AwgnSimulator(SNR=[6.0], modulator="qam", M=16, NSymbs=10000, rep_max=264, seed=5,
                                     batch_size=100, engine=engine, common_random_numbers=True)
    fresh.simulate()
    assert again.results["symbol_errors"][1].get_result() == fresh.results["symbol_errors"][0].get_result()


def test_fading_ofdm_mimo_ia_simulators(engine):
    flat = simulators.FlatFadingSimulator(SNR=[10.0, 30.0], M=64, NSymbs=20000, rep_max=32, batch_size=16,
                                          engine=engine)
    flat.simulate()
    s = flat.results.get_result_values_list("ser")
    assert s[0] > s[1] > 0 and flat.results["ser"][0].num_updates == 32
    ofdm = simulators.OfdmTdlSimulator(SNR=[20.0], rep_max=2000, batch_size=1000, engine=engine)
    ofdm.simulate()
    assert 0.001 < ofdm.results.get_result_values_list("ser")[0] < 0.05
    mimo = simulators.MimoOfdmSimulator(SNR=[15.0, 25.0], rep_max=3000, batch_size=1024, engine=engine)
    mimo.simulate()
    m = mimo.results.get_result_values_list("ser")
    assert m[0] > m[1] and 0.2 < m[1] < 0.35 and mimo.results["num_skipped_reps"][0].get_result() == 0
    assert mimo.results["ber"][1].get_result() < m[1]

    class Stop(simulators.MimoOfdmSimulator):
        def _keep_going(self, params, results, rep):          # reference-style early stop rule
            return results["symbol_errors"][-1].get_result() < 2_000_000
    early = Stop(SNR=[25.0], rep_max=100000, batch_size=512, engine=engine)
    early.simulate()
    assert early.runned_reps[0] < 100000 and early.results["symbol_errors"][0].get_result() >= 2_000_000
    ia = simulators.IaSimulator(SNR=[20.0], rep_max=5000, batch_size=2500, engine=engine)
    ia.simulate()
    assert 0.005 < ia.results.get_result_values_list("ser")[0] < 0.08
    cap = ia.results["sum_capacity"][0]
    assert cap.num_updates == 5000 and 10.0 < cap.get_result() < 25.0 and cap.get_result_var() > 0


def test_exact_early_stop_on_gpu(engine):
    """The stopping rule is applied after every realization by replaying per-realization counts."""
    class Stop(simulators.AwgnSimulator):
        def _keep_going(self, params, results, rep):
            return results["symbol_errors"][-1].get_result() < 3000
    runs = []
    for bs in (8, 1000):
        s = Stop(SNR=[8.0], M=16, NSymbs=1000, rep_max=100000, seed=11, batch_size=bs, engine=engine,
                 exact_early_stop=True)
        s.simulate()
        runs.append((s.runned_reps, s.results["symbol_errors"][0].get_result(), s.results["ser"][0].to_dict()))
    assert runs[0] == runs[1] and 3000 <= runs[0][1] < 3000 + 1000


# ---- SURVEY 8(f).1 as a staged device-resident simulator -------------------------------------------------
F1_KW = dict(mod="qam", M=16, nt=2, nr=2, fft_size=64, cp_size=16, num_used=52, n_ofdm_sym=2, snr_db=14.0,
             Fd=50.0, Ts=1e-6, L=8, tap_powers_dB=(0.0, -4.0, -9.0), tap_delays_samples=(0, 2, 5))


def _f1_sim(dtype, fused=False, **over):
    from pyphysim_amd.simulators import MimoOfdmTdlSimulator
    kw = dict(F1_KW, **over)
    return MimoOfdmTdlSimulator(
        SNR=[kw["snr_db"]], modulator=kw["mod"], M=kw["M"], Nt=kw["nt"], Nr=kw["nr"], fft_size=kw["fft_size"],
        cp_size=kw["cp_size"], num_used_subcarriers=kw["num_used"], num_ofdm_symbols=kw["n_ofdm_sym"], Fd=kw["Fd"],
        Ts=kw["Ts"], L=kw["L"], tap_powers_dB=kw["tap_powers_dB"],
        tap_delays=np.asarray(kw["tap_delays_samples"], dtype=float) * kw["Ts"], seed=77,
        common_random_numbers=True, dtype=dtype, demod="mindist", rep_max=8, batch_size=8, fused=fused)


@pytest.mark.parametrize("over", [dict(), dict(nt=4, nr=4, fft_size=256, num_used=200, n_ofdm_sym=1, M=64, snr_db=24.0),
                                  dict(nt=2, nr=3, num_used=None, tap_powers_dB=(0.0,), tap_delays_samples=(0,))])
def test_mimo_ofdm_tdl_staged_matches_oracle(over):
    """Every realization of the staged HBM-resident chain against the oracle chain on the same Philox draws:
    f64 counts are equal, f32 within a handful of near-tie decisions."""
    from oracle import chains
    kw = dict(F1_KW, **over)
    want_se, want_be = [], []
    for r in range(8):
        o = chains.chain_mimo_ofdm_tdl(chains.PhiloxRng(77, r), **kw)
        want_se.append(o["symbol_errors"])
        want_be.append(o["bit_errors"])
    sim = _f1_sim("f64", **over)
    p = next(iter(sim.params.get_unpacked_params_list()))
    c, se, be = sim._run_batch_detailed(p, 0, 8)
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
    assert c["sym_errors"] == sum(want_se) and c["bit_errors"] == sum(want_be)
    assert c["sym_errors_sq"] == sum(v * v for v in want_se)
    assert c["n_symbols"] == o["num_symbols"] and c["n_realizations"] == 8
    # split batches address the same realizations
    c2, se2, _ = sim._run_batch_detailed(p, 3, 4)
    assert np.array_equal(se2, want_se[3:7])
    sim32 = _f1_sim("f32", **over)
    c32, se32, _ = sim32._run_batch_detailed(p, 0, 8)
    assert abs(int(c32["sym_errors"]) - sum(want_se)) <= 6
    assert np.abs(se32.astype(int) - np.asarray(want_se)).max() <= 3


def test_mimo_ofdm_tdl_simulator_runs_sweep():
    from pyphysim_amd.simulators import MimoOfdmTdlSimulator
    sim = MimoOfdmTdlSimulator(SNR=[5.0, 25.0], M=16, Nt=2, Nr=2, fft_size=64, cp_size=16, num_ofdm_symbols=2,
                               rep_max=512, batch_size=256, seed=5)
    sim.simulate()
    ser = sim.results.get_result_values_list("ser")
    assert ser[0] > ser[1] > 0 or ser[1] == 0
    assert sim.results.get_result_values_list("ser")[0] < 0.7


def test_batched_philox_operators(engine):
    """jakes_taps_philox / awgn_philox / rand_symbols_batch against the oracle draws."""
    from oracle import chains, philox
    from oracle import channels as och
    seed, first, count, L, S, n = 31, 5, 3, 8, 6, 200
    amp = np.linspace(0.3, 1.0, S)
    Ts, Fd = 1e-5, 120.0
    t, _ = och.jakes_time_axis(Ts, Ts, n)
    for dt, tol in (("f64", 1e-10), ("f32", 3e-5)):
        taps = engine.jakes_taps_philox(seed, first, count, L, Fd, Ts, float(t[1] - t[0]), amp, n, dtype=dt).get()
        for r in range(count):
            rng = chains.PhiloxRng(seed, first + r)
            phi, psi = chains._jakes_phases(rng, L, (S,))
            want = och.jakes_samples(phi, psi, Fd, t) * math.sqrt(L) * amp[:, None]
            assert relerr(taps[r], want) <= tol
        x = (np.arange(count * 7 * 11).reshape(count, 7, 11) * (0.5 - 0.25j)).astype(complex)
        y = engine.awgn_philox(x, seed, first, count, 0.37, dtype=dt)
        for r in range(count):
            z = philox.cnormal(seed, first + r, 77, philox.STREAM_NOISE).reshape(7, 11)
            assert relerr(y[r], x[r] + math.sqrt(0.37) * z) <= (1e-12 if dt == "f64" else 1e-5)
    idx = engine.rand_symbols_batch(1001, 64, seed, first, count).get()
    for r in range(count):
        assert np.array_equal(idx[r], philox.symbols(seed, first + r, 1001, 64))


@pytest.mark.parametrize("over", [dict(), dict(cp_size=4), dict(n_ofdm_sym=3, tap_delays_samples=(0, 7, 19), cp_size=8),
                                  dict(nt=4, nr=4, fft_size=256, num_used=200, n_ofdm_sym=1, M=64, snr_db=24.0),
                                  dict(nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, M=64,
                                       snr_db=25.0, Fd=10.0, Ts=1.0 / (15e3 * 1024),
                                       tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0),
                                       tap_delays_samples=(0, 1, 2, 3, 4)),
                                  dict(Fd=70.0, Ts=1e-5),
                                  dict(fft_size=128, num_used=100, cp_size=9),
                                  dict(nt=4, nr=4, fft_size=512, num_used=300, cp_size=36, n_ofdm_sym=1, M=64, snr_db=24.0),
                                  dict(fft_size=2048, num_used=1200, cp_size=144, n_ofdm_sym=1, Fd=5.0)])
def test_mimo_ofdm_tdl_fused_matches_oracle(over):
    """The fused kernel (polynomial tap model) against the oracle chain on the same Philox draws: equal
    per-realization counts in f64 (incl. inter-symbol interference through the CP, cp < max delay, and a
    Doppler high enough for a higher polynomial order), a few near-tie flips in f32; equal to the staged chain."""
    from oracle import chains
    kw = dict(F1_KW, **over)
    n = 4 if kw["fft_size"] >= 1024 else 8
    want_se, want_be = [], []
    for r in range(n):
        o = chains.chain_mimo_ofdm_tdl(chains.PhiloxRng(77, r), **kw)
        want_se.append(o["symbol_errors"])
        want_be.append(o["bit_errors"])
    sim = _f1_sim("f64", fused=True, **over)
    p = next(iter(sim.params.get_unpacked_params_list()))
    c, se, be = sim._run_batch_detailed(p, 0, n)
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be)
    assert c["sym_errors"] == sum(want_se) and c["n_symbols"] == o["num_symbols"] and c["n_realizations"] == n
    c2, se2, _ = sim._run_batch_detailed(p, 1, n - 2)
    assert np.array_equal(se2, want_se[1:n - 1])
    sim32 = _f1_sim("f32", fused=True, **over)
    c32, se32, _ = sim32._run_batch_detailed(p, 0, n)
    assert np.abs(se32.astype(int) - np.asarray(want_se)).max() <= 3
    st32 = _f1_sim("f32", fused=False, **over)
    _, se_st, _ = st32._run_batch_detailed(p, 0, n)
    assert np.abs(se32.astype(int) - se_st.astype(int)).max() <= 3


def test_mimo_ofdm_tdl_fused_envelope():
    """Beyond the tap model's Doppler envelope the C ABI reports MCLE_E_UNSUPPORTED; 'auto' falls back to
    the staged chain (same counts as the oracle), fused=True raises."""
    from oracle import chains
    from pyphysim_amd._lib import McleUnsupported
    over = dict(Fd=30000.0, Ts=1e-5)
    kw = dict(F1_KW, **over)
    want = [chains.chain_mimo_ofdm_tdl(chains.PhiloxRng(77, r), **kw)["symbol_errors"] for r in range(4)]
    sim = _f1_sim("f64", fused="auto", **over)
    p = next(iter(sim.params.get_unpacked_params_list()))
    _, se, _ = sim._run_batch_detailed(p, 0, 4)
    assert np.array_equal(se, want)
    with pytest.raises(McleUnsupported):
        _f1_sim("f64", fused=True, **over)._run_batch_detailed(p, 0, 4)


def test_ia_simulator_iterative(engine):
    """IaSimulator with an iterative solver: same Result surface as the reference app incl.
    'ia_runned_iterations' (apps/ia/simulate_ia.py:185-189); max SINR beats closed form at low SNR."""
    out = {}
    for solver in ("closed_form", "max_sinr", "alt_min"):
        sim = simulators.IaSimulator(SNR=[0.0, 20.0], M=4, modulator="psk", NSymbs=100, rep_max=4000, batch_size=4000,
                                     seed=3, engine=engine, solver=solver, max_iterations=30)
        sim.simulate()
        out[solver] = sim.results
        assert "sum_capacity" in sim.results.get_result_names()
        assert ("ia_runned_iterations" in sim.results.get_result_names()) == (solver != "closed_form")
    its = out["alt_min"].get_result_values_list("ia_runned_iterations")
    assert all(1.0 <= v <= 30.0 for v in its)
    cap = {k: v.get_result_values_list("sum_capacity") for k, v in out.items()}
    assert cap["max_sinr"][0] > cap["closed_form"][0]            # noise-aware filters win at 0 dB
    assert abs(cap["alt_min"][1] - cap["closed_form"][1]) < 0.15 * cap["closed_form"][1]   # both align at 20 dB


def test_mimo_simulator_schemes(engine):
    """MimoSimulator = the reference's MIMO application (apps/mimo/simulate_mimo.py) for its six schemes:
    diversity orders show in the SER ordering, SVD / GMD / Blast all resolve Nt layers."""
    ser = {}
    for scheme, nt, nr in (("blast", 2, 2), ("alamouti", 2, 2), ("mrc", 1, 2), ("mrt", 2, 1), ("svd", 2, 2),
                           ("gmd", 2, 2)):
        sim = simulators.MimoSimulator(SNR=[10.0], scheme=scheme, M=4, modulator="psk", Nt=nt, Nr=nr, NSymbs=100,
                                       rep_max=20000, batch_size=20000, seed=9, engine=engine)
        sim.simulate()
        ser[scheme] = sim.results.get_result_values_list("ser")[0]
        layers = 1 if scheme in ("alamouti", "mrt") else nt
        assert sim.results.get_result_values_list("num_symbols")[0] == 20000 * 100 * layers
    assert ser["alamouti"] < ser["mrc"] < ser["blast"]        # diversity 4 > 2 > 1 (ZF 2x2)
    assert ser["alamouti"] < ser["mrt"]
    assert 0 < ser["gmd"] and 0 < ser["svd"] < 0.5
    with pytest.raises(ValueError):
        simulators.MimoSimulator(SNR=[10.0], scheme="stbc")
