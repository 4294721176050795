"""GPU: the ready-made simulators (BatchedSimulationRunner + fused pipelines) end to end through the
kept SimulationRunner / SimulationResults surface."""
import numpy as np
import pytest

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
