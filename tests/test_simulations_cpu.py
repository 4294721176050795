"""CPU: host logic of the kept pyphysim.simulations surface (Result, SimulationResults,
SimulationParameters, SimulationRunner, BatchedSimulationRunner) against known answers minted
from the reference (tests/golden/framework.json) and the semantics its own tests pin
(reference tests/simulations_package_test.py:1272-1845, 2364-2703, 2908+)."""
import json
import os

import numpy as np
import pytest

from pyphysim_amd.simulations import (BatchedSimulationRunner, Result, SimulationParameters, SimulationResults,
                                      SimulationRunner, SkipThisOne, calc_confidence_interval)

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "framework.json")))


def test_ratio_result_matches_reference():
    g = G["ratio"]
    r = Result("ser", Result.RATIOTYPE)
    for v, t in zip(g["values"], g["totals"]):
        r.update(v, t)
    assert [r._value, r._total, r._result_sum, r._result_squared_sum, r.num_updates] == g["state"]
    assert r.get_result() == g["result"] and r.get_result_mean() == g["mean"] and r.get_result_var() == g["var"]
    assert list(r.get_confidence_interval(95)) == g["ci95"] and list(r.get_confidence_interval(99)) == g["ci99"]
    with pytest.raises(ValueError):
        r.update(3)
    assert Result("x", Result.RATIOTYPE).get_result() == "Nothing yet"
    assert r.type_name == "RATIOTYPE" and r.type_code == 1


def test_sum_merge_choice_misc():
    g = G["sum_merge"]
    a, b = Result("errs", Result.SUMTYPE), Result("errs", Result.SUMTYPE)
    for v in g["a"]:
        a.update(v)
    for v in g["b"]:
        b.update(v)
    a.merge(b)
    assert [a._value, a._total, a._result_sum, a._result_squared_sum, a.num_updates] == g["state"]
    assert a.get_result_var() == g["var"]
    c = Result.create("choice", Result.CHOICETYPE, G["choice"]["updates"][0], 4)
    for v in G["choice"]["updates"][1:]:
        c.update(v)
    assert [float(x) for x in c.get_result()] == G["choice"]["result"]
    with pytest.raises(RuntimeError):
        Result.create("c", Result.CHOICETYPE, 1)
    m = Result.create("m", Result.MISCTYPE, "a")
    m.merge(Result.create("m", Result.MISCTYPE, "b"))
    assert m.get_result() == "b"
    with pytest.raises(RuntimeError):
        m.get_confidence_interval()
    with pytest.raises(AssertionError):
        a.merge(Result("other", Result.SUMTYPE))
    assert Result.from_dict(a.to_dict()) == a and Result.from_json(a.to_json()) == a


def test_from_counters_equals_sequential_updates():
    rs = np.random.RandomState(4)
    e = rs.randint(0, 300, 1000)
    units = 4096
    seq_ratio, seq_sum = Result("ser", Result.RATIOTYPE), Result("symbol_errors", Result.SUMTYPE)
    for v in e:
        seq_ratio.update(int(v), units)
        seq_sum.update(int(v))
    fr = Result.from_counters("ser", Result.RATIOTYPE, int(e.sum()), int((e.astype(np.int64) ** 2).sum()), units, e.size)
    fs = Result.from_counters("symbol_errors", Result.SUMTYPE, int(e.sum()), int((e.astype(np.int64) ** 2).sum()), 1,
                              e.size)
    assert (fr._value, fr._total, fr.num_updates) == (seq_ratio._value, seq_ratio._total, seq_ratio.num_updates)
    assert fr.get_result() == seq_ratio.get_result()
    assert abs(fr._result_sum - seq_ratio._result_sum) < 1e-12 * seq_ratio._result_sum
    assert abs(fr.get_result_var() - seq_ratio.get_result_var()) < 1e-12
    assert (fs._value, fs._result_sum, fs._result_squared_sum) == (seq_sum._value, seq_sum._result_sum,
                                                                   seq_sum._result_squared_sum)
    lo, hi = fr.get_confidence_interval(95)
    lo2, hi2 = seq_ratio.get_confidence_interval(95)
    assert abs(lo - lo2) < 1e-13 and abs(hi - hi2) < 1e-13
    assert calc_confidence_interval(1.0, 2.0, 4, 95) == (1.0 - 1.96, 1.0 + 1.96)


def test_parameters_unpacking_matches_reference():
    g = G["params"]
    p = SimulationParameters.create({"SNR": np.array([0, 5, 10]), "M": [4, 16], "alpha": 0.5, "name": "x"})
    p.set_unpack_parameter("SNR")
    p.set_unpack_parameter("M")
    lst = p.get_unpacked_params_list()
    assert [[int(q["M"]), int(q["SNR"])] for q in lst] == g["order"]
    assert p.get_num_unpacked_variations() == g["n"] and [q.unpack_index for q in lst] == g["unpack_index"]
    assert [int(i) for i in p.get_pack_indexes({"SNR": 5})] == g["idx_snr5"]
    assert [int(i) for i in p.get_pack_indexes({"M": 16})] == g["idx_m16"]
    assert [int(i) for i in p.get_pack_indexes({"M": 16, "SNR": 10})] == g["idx_both"]
    assert lst[2].get_num_unpacked_variations() == 6 and lst[0]["alpha"] == 0.5
    with pytest.raises(ValueError):
        p.set_unpack_parameter("alpha")
    with pytest.raises(ValueError):
        p.set_unpack_parameter("nope")
    assert p.unpacked_parameters == ["M", "SNR"] and sorted(p.fixed_parameters) == ["alpha", "name"]
    q = SimulationParameters.from_json(p.to_json())
    assert q == p
    other = SimulationParameters.create({"SNR": np.array([0, 5, 11]), "M": [4, 16], "alpha": 0.5, "name": "x"})
    other.set_unpack_parameter("SNR")
    other.set_unpack_parameter("M")
    assert other != p


class _Dummy(SimulationRunner):
    """The reference's _DummyRunner (tests/simulations_package_test.py:2364-2400): deterministic."""

    def __init__(self):
        super().__init__(read_command_line_args=False)
        self.rep_max = 2
        self.params.add("SNR", np.array([0.0, 5.0, 10.0, 15.0, 20.0]))
        self.params.set_unpack_parameter("SNR")
        self.params.add("bias", 1.3)
        self.params.add("extra", np.array([2.2, 4.1]))
        self.params.set_unpack_parameter("extra")

    def _run_simulation(self, current_params):
        value = 1.2 * current_params["SNR"] + current_params["bias"] + current_params["extra"]
        res = SimulationResults()
        res.add_new_result("lala", Result.RATIOTYPE, value, 1)
        return res


def test_runner_serial_loop(tmp_path):
    d = _Dummy()
    d.simulate()
    lala = d.results.get_result_values_list("lala")
    want = [1.2 * snr + 1.3 + ex for snr in (0.0, 5.0, 10.0, 15.0, 20.0) for ex in (2.2, 4.1)]
    np.testing.assert_array_almost_equal(lala, want)           # sorted names: SNR outer, extra inner
    assert d.runned_reps == [2] * 10
    assert set(d.results.get_result_names()) == {"lala", "elapsed_time", "num_skipped_reps"}
    assert d.results.get_result_values_list("lala", {"SNR": 10.0}) == pytest.approx([1.2 * 10 + 1.3 + 2.2, 1.2 * 10 + 1.3 + 4.1])
    assert all(r.num_updates == 2 for r in d.results["lala"])
    # save / load round trips (pickle and json)
    d.results.set_parameters(d.params)
    for ext in ("pickle", "json"):
        name = d.results.save_to_file(str(tmp_path / ("res." + ext)))
        back = SimulationResults.load_from_file(name)
        np.testing.assert_array_almost_equal(back.get_result_values_list("lala"), want)


class _Skipper(SimulationRunner):
    def __init__(self):
        super().__init__(read_command_line_args=False)
        self.rep_max = 5
        self.calls = 0

    def _run_simulation(self, current_params):
        self.calls += 1
        if self.calls in (2, 4):
            raise SkipThisOne("nope")
        res = SimulationResults()
        res.add_new_result("v", Result.SUMTYPE, 1)
        return res


def test_skip_this_one_and_keep_going():
    s = _Skipper()
    s.simulate()
    assert s.results["v"][0].get_result() == 5 and s.results["num_skipped_reps"][0].get_result() == 2
    assert s.calls == 7

    class Stopper(_Skipper):
        def _keep_going(self, p, res, rep):
            return res["v"][-1].get_result() < 3
    t = Stopper()
    t.calls = 100
    t.simulate()
    assert t.results["v"][0].get_result() == 3 and t.runned_reps == [3]


def test_partial_results_resume_and_mismatch(tmp_path):
    d = _Dummy()
    d.set_results_filename(str(tmp_path / "dummy"))
    d.simulate()
    files = sorted(os.listdir(tmp_path / "partial_results"))
    assert len(files) == 10 and files[0] == "dummy_unpack_00.pickle"
    first = d.results.get_result_values_list("lala")
    d2 = _Dummy()
    d2.rep_max = 4                                    # resume: 2 more repetitions per variation
    d2.set_results_filename(str(tmp_path / "dummy"))
    calls = []
    orig = d2._run_simulation
    d2._run_simulation = lambda p: (calls.append(1), orig(p))[1]
    d2.simulate()
    assert len(calls) == 20 and d2.runned_reps == [4] * 10
    np.testing.assert_array_almost_equal(d2.results.get_result_values_list("lala"), first)
    d3 = _Dummy()
    d3.params.add("bias", 9.9)
    d3.set_results_filename(str(tmp_path / "dummy"))
    with pytest.raises(ValueError):
        d3.simulate()
    d4 = _Dummy()
    with pytest.raises(RuntimeError):
        d4.simulate(param_variation_index=3)          # needs a results filename (runner.py:1658)


def fake_counters(first, count, units=100, bits=2):
    """Deterministic per-index 'errors' so that sharding / batching invariance can be checked."""
    r = np.arange(first, first + count, dtype=np.uint64)
    e = ((r * np.uint64(2654435761)) % np.uint64(97)).astype(np.int64)
    skip = (r % np.uint64(50) == np.uint64(49))
    b = 2 * e[~skip]
    e = e[~skip]
    return dict(n_realizations=int(e.size), n_skipped=int(skip.sum()), sym_errors=int(e.sum()),
                sym_errors_sq=int((e * e).sum()), bit_errors=int(b.sum()), bit_errors_sq=int((b * b).sum()),
                n_symbols=units, n_bits=units * bits)


class FakeBatched(BatchedSimulationRunner):
    def __init__(self, batch_size, rep_max=1000):
        super().__init__(batch_size=batch_size)
        self.rep_max = rep_max
        self.params.add("SNR", np.array([0.0, 10.0]))
        self.params.set_unpack_parameter("SNR")
        self.calls = []

    def _run_batch(self, current_parameters, first_rep, count):
        self.calls.append((first_rep, count))
        return fake_counters(first_rep, count)


class FakeBatchedExtra(FakeBatched):
    """Carries a float side sum next to the counters, like IaSimulator's sum capacity."""
    EXTRA_KEYS = ("cap", "cap_sq", "its")
    EXTRA_INT_KEYS = ("its",)

    def _run_batch(self, current_parameters, first_rep, count):
        c = super()._run_batch(current_parameters, first_rep, count)
        idx = np.arange(first_rep, first_rep + count)
        good = idx % 50 != 49                      # fake_counters skips every 50th index
        vals = 1.0 + (idx[good] % 7) * 0.125
        c.update(cap=float(vals.sum()), cap_sq=float((vals ** 2).sum()), its=int((idx[good] % 5).sum()))
        return c

    def _results_from_counters(self, current_parameters, c):
        res = super()._results_from_counters(current_parameters, c)
        n = max(int(c["n_realizations"]), 1)
        res.add_result(Result.from_batch("cap", Result.RATIOTYPE, c.get("cap", 0.0), n, c.get("cap", 0.0),
                                         c.get("cap_sq", 0.0), n))
        return res


class AllSkipped(BatchedSimulationRunner):
    def __init__(self):
        super().__init__(batch_size=10)
        self.rep_max = 100
        self.params.add("SNR", np.array([0.0]))
        self.params.set_unpack_parameter("SNR")

    def _run_batch(self, current_parameters, first_rep, count):
        return dict(n_realizations=0, n_skipped=count, sym_errors=0, sym_errors_sq=0, bit_errors=0, bit_errors_sq=0,
                    n_symbols=8, n_bits=16)


def test_a_configuration_that_skips_everything_is_an_error_not_a_hang():
    sim = AllSkipped()
    sim.max_all_skipped_batches = 5
    with pytest.raises(RuntimeError, match="skipped"):
        sim.simulate()


def test_extra_sums_are_batch_size_invariant_and_resume(tmp_path):
    outs = []
    for bs in (1000, 37):
        f = FakeBatchedExtra(bs)
        f.simulate()
        outs.append([r.to_dict() for r in f.results["cap"]])
    assert outs[0][0]["num_updates"] == 1000
    for a, b in zip(outs[0], outs[1]):
        assert abs(a["value"] - b["value"]) < 1e-9 and a["total"] == b["total"]
    g = FakeBatchedExtra(100, rep_max=300)
    g.set_results_filename(str(tmp_path / "fx"))
    g.simulate()
    h = FakeBatchedExtra(100, rep_max=1000)
    h.set_results_filename(str(tmp_path / "fx"))
    h.simulate()
    for a, b in zip(outs[0], [r.to_dict() for r in h.results["cap"]]):
        assert abs(a["value"] - b["value"]) < 1e-9 and a["num_updates"] == b["num_updates"]


def test_partial_saves_follow_realizations_not_batches(tmp_path, monkeypatch):
    """with a results file set, a resume point is written every `partial_save_every_reps` realizations,
    not after every GPU batch."""
    f = FakeBatched(10, rep_max=1000)
    f.partial_save_every_reps = 400
    f.set_results_filename(str(tmp_path / "cad"))
    saves = []
    orig = f._save_partial
    monkeypatch.setattr(f, "_save_partial", lambda *a, **k: (saves.append(a[0]), orig(*a, **k))[1])
    f.simulate()
    per_variation = len(saves) / 2
    assert 2 <= per_variation <= 5, saves            # ~1000 / 400 mid-run saves + the final one, not 100


def test_batched_runner_is_batch_size_invariant(tmp_path):
    outs = []
    for bs in (1000, 64, 7):
        f = FakeBatched(bs)
        f.simulate()
        outs.append({n: [r.to_dict() for r in f.results[n]] for n in ("ser", "ber", "symbol_errors", "num_symbols",
                                                                       "bit_errors", "num_bits")})
        assert f.runned_reps == [1000, 1000]
        assert f.results["num_skipped_reps"][0].get_result() >= 20       # every 50th index is skipped
        assert f.results["ser"][0].num_updates == 1000
    assert outs[0] == outs[1] == outs[2]
    # replacement of skipped realizations: the index range grows past rep_max
    f = FakeBatched(100)
    f.simulate()
    assert max(a + b for a, b in f.calls) > 1000
    # the Result set is the reference simulators' (simulate_psk.py:90-112)
    assert set(f.results.get_result_names()) == {"symbol_errors", "num_symbols", "bit_errors", "num_bits", "ber",
                                                 "ser", "elapsed_time", "num_skipped_reps"}
    # resume from partial results continues at the next unseen index
    g = FakeBatched(100, rep_max=300)
    g.set_results_filename(str(tmp_path / "fb"))
    g.simulate()
    h = FakeBatched(100, rep_max=1000)
    h.set_results_filename(str(tmp_path / "fb"))
    h.simulate()
    assert h.calls[0][0] == max(a + b for a, b in g.calls[:len(g.calls) // 2])
    assert [r.to_dict() for r in h.results["ser"]] == outs[0]["ser"]

    class EarlyStop(FakeBatched):
        def _keep_going(self, p, res, rep):
            return res["symbol_errors"][-1].get_result() < 5000
    e = EarlyStop(50)
    e.simulate()
    assert all(r < 1000 for r in e.runned_reps) and all(v >= 5000 for v in e.results.get_result_values_list("symbol_errors"))


def test_shard_range_partitions():
    for first, count, world in ((0, 10, 3), (5, 1, 4), (7, 4096, 8), (0, 0, 2)):
        parts = [BatchedSimulationRunner.shard_range(first, count, r, world) for r in range(world)]
        assert sum(c for _, c in parts) == count
        pos = first
        for lo, c in parts:
            assert lo == pos
            pos += c


def test_reference_result_archives_interoperate(tmp_path):
    """an archive written by the reference (tests/golden/reference_results.pickle,
    pickle protocol 2) loads without the reference installed, and what compat.save_for_reference
    writes names the reference's classes (the reverse direction is asserted against the reference
    itself in oracle/make_golden.py)."""
    from pyphysim_amd.simulations import compat
    path = os.path.join(os.path.dirname(__file__), "golden", "reference_results.pickle")
    res = compat.load_reference_results(path)
    g = G["archive"]
    assert res.get_result_values_list("ser") == g["ser"]
    assert res.get_result_values_list("symbol_errors") == g["symbol_errors"]
    assert res.get_result_values_list("ser", {"M": 16}) == g["ser_m16"]
    assert res.params.get_num_unpacked_variations() == 6 and res.runned_reps == [5] * 6
    lo, hi = res["ser"][0].get_confidence_interval(95)
    assert lo < res["ser"][0].get_result_mean() < hi
    out = compat.save_for_reference(res, str(tmp_path / "w.pickle"))
    raw = open(out, "rb").read()
    assert b"pyphysim.simulations.results\nSimulationResults" in raw and b"pyphysim_amd" not in raw
    again = compat.load_reference_results(out)
    assert again.get_result_values_list("ser") == g["ser"] and again.params == res.params


def test_exact_early_stop_replays_every_realization():
    """With exact_early_stop the stopping rule sees the state after EVERY realization, like the
    reference's serial loop (runner.py:1491-1517): the result equals a batch-size-1 run."""
    class Exact(FakeBatched):
        def _run_batch_detailed(self, p, first, count):
            c = fake_counters(first, count)
            r = np.arange(first, first + count, dtype=np.uint64)
            e = ((r * np.uint64(2654435761)) % np.uint64(97)).astype(np.int64)
            skip = (r % np.uint64(50) == np.uint64(49))
            se = np.where(skip, 0xFFFFFFFF, e).astype(np.uint32)
            be = np.where(skip, 0xFFFFFFFF, 2 * e).astype(np.uint32)
            return c, se, be

        def _keep_going(self, p, res, rep):
            return res["symbol_errors"][-1].get_result() < 5000
    outs = []
    for bs in (1, 37, 1000):
        e = Exact(bs)
        e.exact_early_stop = True
        e.simulate()
        outs.append(([r.to_dict() for r in e.results["ser"]], e.runned_reps))
    assert outs[0] == outs[1] == outs[2]
    errs = [d["value"] for d in outs[0][0]]
    assert all(5000 <= v < 5000 + 97 for v in errs)          # stopped at the first realization crossing the rule


def test_reference_json_archives_interoperate(tmp_path):
    """the reference's JSON archives (results.py:1475-1486; arrays and sets as the tagged dicts
    of util/serialize.py) load here, and what we write uses the same encoding (the reference reading it back is
    asserted in oracle/make_golden.py)."""
    import json
    from pyphysim_amd.simulations import SimulationParameters, SimulationResults
    path = os.path.join(os.path.dirname(__file__), "golden", "reference_results.json")
    res = SimulationResults.load_from_file(path)
    g = G["archive"]
    assert res.get_result_values_list("ser") == g["ser"]
    assert res.get_result_values_list("symbol_errors") == g["symbol_errors"]
    assert res.get_result_values_list("ser", {"M": 16}) == g["ser_m16"]
    assert res.params.get_num_unpacked_variations() == 6 and res.runned_reps == [5] * 6
    assert isinstance(res.params["SNR"], np.ndarray)
    out = res.save_to_file(str(tmp_path / "w.json"))
    doc = json.loads(open(out).read())
    want = json.loads(open(path).read())
    assert doc["params"]["parameters"].keys() == want["params"]["parameters"].keys()
    for k, v in want["params"]["parameters"].items():
        if isinstance(v, dict):
            assert doc["params"]["parameters"][k]["_is_numpy_array"] is True
            assert doc["params"]["parameters"][k]["data"] == v["data"] and doc["params"]["parameters"][k]["dtype"] == v["dtype"]
    assert doc["params"]["unpacked_parameters_set"]["_is_set"] is True
    assert sorted(doc["params"]["unpacked_parameters_set"]["data"]) == sorted(want["params"]["unpacked_parameters_set"]["data"])
    assert doc["results"]["ser"] == want["results"]["ser"]
    again = SimulationResults.load_from_file(out)
    assert again.get_result_values_list("ser") == g["ser"] and again.params == res.params
    # SimulationParameters on its own
    p = SimulationParameters.create({"a": np.arange(3.0), "b": 2, "c": np.array([1, 5])})
    p.set_unpack_parameter("a")
    q = SimulationParameters.from_json(p.to_json())
    assert q == p and q.get_num_unpacked_variations() == 3 and q["c"].dtype == p["c"].dtype


def test_range_representations_and_file_names():
    """Result file names embed the parameter arrays as range expressions (util/misc.py:911-1115);
    known answers from the reference."""
    from pyphysim_amd.simulations.parameters import get_mixed_range_representation, replace_dict_values
    for case in G["range_representations"]:
        arr = np.array(case["data"], dtype=int if case["is_int"] else float)
        assert get_mixed_range_representation(arr, case["filename_mode"]) == case["text"], case
    r = G["replace_dict_values"]
    assert replace_dict_values(r["name"], {"SNR": np.array(r["SNR"]), "M": r["M"], "tag": r["tag"]}, True) == r["text"]


def test_combine_and_split_workflows(tmp_path):
    """bin/combine_results.py and bin/split_into_partial_results.py on archives written by the reference;
    the combined values are the reference's own combine_simulation_results output."""
    from pyphysim_amd.simulations import (SimulationParameters, SimulationResults, SimulationRunner,
                                          combine_simulation_parameters, combine_simulation_results,
                                          get_partial_results_filename, tools)
    gold = os.path.join(os.path.dirname(__file__), "golden")
    fa, fb = os.path.join(gold, "reference_results_a.json"), os.path.join(gold, "reference_results_b.json")
    out = tools.combine_main([fa, fb, str(tmp_path / "u.json")])
    u = SimulationResults.load_from_file(out)
    g = G["combined"]
    assert list(u.params["SNR"]) == g["SNR"]
    assert u.get_result_values_list("ser") == g["ser"]
    assert u.get_result_values_list("symbol_errors") == g["symbol_errors"]
    assert [r.num_updates for r in u["ser"]] == g["num_updates"]
    with pytest.raises(RuntimeError):
        tools.combine_main([fa, fb, fa])
    a, b = SimulationResults.load_from_file(fa), SimulationResults.load_from_file(fb)
    b2 = SimulationResults.load_from_file(fb)
    b2._results.pop("ser")
    with pytest.raises(RuntimeError):
        combine_simulation_results(a, b2)
    p1 = SimulationParameters.create({"SNR": np.array([0., 5.]), "M": 4})
    p2 = SimulationParameters.create({"SNR": np.array([0., 5.]), "M": 8})
    for p in (p1, p2):
        p.set_unpack_parameter("SNR")
    with pytest.raises(RuntimeError):
        combine_simulation_parameters(p1, p2)
    # split: one partial file per variation, named like the runner names them, loadable and consistent
    u.original_filename = str(tmp_path / "sweep_{M}")
    u.runned_reps = [r.num_updates for r in u["ser"]]
    whole = str(tmp_path / "whole.json")
    with open(whole, "w") as fh:
        fh.write(u.to_json())
    files = tools.split_main([whole, str(tmp_path / "partial_results")])
    assert len(files) == len(g["SNR"])
    unpacked = u.params.get_unpacked_params_list()
    for i, f in enumerate(files):
        assert os.path.basename(f) == os.path.basename(get_partial_results_filename("sweep_16", unpacked[i]))
        part = SimulationResults.load_from_file(f)
        assert part["ser"][0].get_result() == g["ser"][i] and part.current_rep == u.runned_reps[i]
    assert get_partial_results_filename("base", unpacked[2], "dir") == os.path.join("dir", "base_unpack_2.pickle")


class _FakeAsync:
    def __init__(self, values):
        self._values, self.waited = values, False

    def wait(self):
        self.waited = True

    def get(self):
        return self._values


class _FakeView:
    """An ipyparallel-style view: map(func, *iterables, block=False) -> object with wait() / get()."""

    def __init__(self):
        self.calls = 0

    def map(self, func, *iterables, block=False):
        assert block is False
        out = []
        for args in zip(*iterables):
            self.calls += 1
            out.append(func(*args))
        return _FakeAsync(out)


def test_simulate_in_parallel_mirrors_the_task_parallel_mode(tmp_path):
    """runner.py:1774-1886: one variation per engine of the view, results appended in variation order; without a view
    the variations run through simulate()."""
    serial = _Dummy()
    serial.simulate()
    view = _FakeView()
    par = _Dummy()
    par.set_results_filename(str(tmp_path / "par"))
    par.simulate_in_parallel(view)
    assert view.calls == par.params.get_num_unpacked_variations()
    assert par.runned_reps == serial.runned_reps
    for name in serial.results.get_result_names():
        if name != "elapsed_time":
            assert [r.to_dict() for r in par.results[name]] == [r.to_dict() for r in serial.results[name]]
    assert (tmp_path / "par.pickle").exists()
    par.wait_parallel_simulation()                      # a second call is a no-op
    lazy = _Dummy()
    lazy.simulate_in_parallel(_FakeView(), wait=False)
    assert lazy.runned_reps == []
    lazy.wait_parallel_simulation()
    assert lazy.runned_reps == serial.runned_reps
    plain = _Dummy()
    plain.simulate_in_parallel()
    assert plain.runned_reps == serial.runned_reps


def test_private_dict_serialisation_names_of_the_reference():
    """the reference spells the dict round trip `_to_dict` / `_from_dict` (simulations/results.py:725,747,
    1361,1408; simulations/parameters.py:942,963); a subclass written against pyphysim calls those names."""
    p = SimulationParameters.create({"SNR": np.array([0.0, 5.0, 10.0]), "M": 16})
    p.set_unpack_parameter("SNR")
    assert SimulationParameters._from_dict(p._to_dict()) == p
    res = SimulationResults()
    res.set_parameters(p)
    for v in (3, 5, 7):
        r = Result.create("errs", Result.RATIOTYPE, v, 100)
        assert Result._from_dict(r._to_dict()) == r
        res.append_result(r)
    back = SimulationResults._from_dict(res._to_dict())
    assert back == res and back.params == p
    assert [r.get_result() for r in back["errs"]] == [0.03, 0.05, 0.07]

    class Mine(SimulationResults):            # the calling pattern of a user subclass
        def dump(self):
            return self._to_dict()
    m = Mine()
    m.set_parameters(p)
    m.add_new_result("x", Result.SUMTYPE, 4)
    assert SimulationResults._from_dict(m.dump())["x"][0].get_result() == 4
