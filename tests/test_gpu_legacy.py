"""GPU: same-seed parity.  The device replays NumPy's legacy MT19937 stream for
np.random.seed(seed) (csrc/kernels_legacy.hip) and the per-operator HIP kernels run the
reference's chains on those draws; the raw draws and the final error counts must equal what the
REFERENCE produced for the same seed (tests/golden/*.npz, minted by oracle/make_golden.py)."""
import numpy as np
import pytest

from helpers import golden_cases
from pyphysim_amd import legacy

pytestmark = pytest.mark.gpu


def test_legacy_stream_matches_numpy(engine):
    """Draw-level check against NumPy itself (the generator the reference uses)."""
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        prog = [("randint", 37, 64), ("randn", 5), ("rand", 6), ("randint", 3, 2), ("randn", 1001), ("randn", 4)]
        ints, dbls = engine.legacy_draws(prog, seed, 0, 3)
        for r in range(3):
            rs = np.random.RandomState((seed + r) & 0xFFFFFFFF)
            a = rs.randint(0, 64, 37)
            g1 = rs.randn(5)
            u = rs.rand(6)
            b = rs.randint(0, 2, 3)
            g2 = rs.randn(1001)
            g3 = rs.randn(4)
            assert np.array_equal(ints.get()[r], np.concatenate([a, b]))
            want = np.concatenate([g1, u, g2, g3])
            assert np.max(np.abs(dbls.get()[r] - want)) <= 4e-16 * np.max(np.abs(want))
            assert np.array_equal(dbls.get()[r][5:11], u)             # uniforms are exact


def _check(name, runner, keymap=None):
    for kw, reals in golden_cases(name):
        seeds = [int(g["seed"]) for g in reals]
        assert seeds == list(range(seeds[0], seeds[0] + len(seeds)))
        out = runner(seeds[0], len(seeds), kw)
        se, be = out[-2], out[-1]
        assert [int(v) for v in se] == [int(g["symbol_errors"]) for g in reals], (name, kw)
        assert [int(v) for v in be] == [int(g["bit_errors"]) for g in reals], (name, kw)


def test_awgn_same_seed_as_reference(engine):
    _check("c1_awgn", lambda s, n, kw: legacy.run_awgn(engine, s, 0, n, **kw))


def test_flat_jakes_same_seed_as_reference(engine):
    _check("c2_flat_jakes", lambda s, n, kw: legacy.run_flat_jakes(engine, s, 0, n, **kw))


def test_flat_rayleigh_same_seed_as_reference(engine):
    """a7: both of the reference's statements of i.i.d. Rayleigh fading (channel classes / notebook cell 8)."""
    _check("c2b_flat_rayleigh", lambda s, n, kw: legacy.run_flat_rayleigh(engine, s, 0, n, **kw))


def test_ofdm_tdl_same_seed_as_reference(engine):
    _check("c3_ofdm_tdl", lambda s, n, kw: legacy.run_ofdm_tdl(engine, s, 0, n, **kw))


def test_mimo_ofdm_same_seed_as_reference(engine):
    _check("c4_mimo_ofdm", lambda s, n, kw: legacy.run_mimo_ofdm(engine, s, 0, n, **kw))


def test_ia_same_seed_as_reference(engine):
    """Config 5: three RandomStates (channel, noise, data) replayed on the device."""
    _check("c5_ia", lambda s, n, kw: legacy.run_ia(engine, s, 0, n, **kw))


def test_ia_iterative_same_seed_as_reference(engine):
    """plus the iterative solvers' own RandomState for the initial precoders; error counts,
    iteration counts and capacities equal the reference run with the same seeds."""
    for kw, reals in golden_cases("f3_ia_iterative"):
        seeds = [int(g["seed"]) for g in reals]
        cnt, sol, se, be = legacy.run_ia(engine, seeds[0], 0, len(seeds), **kw)
        assert [int(v) for v in se] == [int(g["symbol_errors"]) for g in reals], kw
        assert [int(v) for v in be] == [int(g["bit_errors"]) for g in reals], kw
        assert [int(v) for v in sol["iterations"]] == [int(g["runned_iterations"]) for g in reals], kw
        assert np.allclose(sol["capacity"], [float(g["sum_capacity"]) for g in reals], rtol=0, atol=1e-6)


def test_mimo_ofdm_tdl_same_seed_as_reference(engine):
    """SURVEY 8(f).1 (frequency-selective MIMO-OFDM) under np.random.seed."""
    _check("f1_mimo_ofdm_tdl", lambda s, n, kw: legacy.run_mimo_ofdm_tdl(engine, s, 0, n, **kw))


def test_mimo_schemes_same_seed_as_reference(engine):
    """apps/mimo/simulate_mimo.py with Blast / MRC / MRT / Alamouti under np.random.seed: the reference's counts."""
    seen = set()
    for kw, reals in golden_cases("f5_mimo_schemes"):
        if kw["scheme"] in ("svd", "gmd"):
            continue                              # LAPACK's singular-vector phases are not reproducible
        seen.add(kw["scheme"])
        seeds = [int(g["seed"]) for g in reals]
        se, be = legacy.run_mimo_scheme(engine, seeds[0], 0, len(seeds), **kw)
        assert [int(v) for v in se] == [int(g["symbol_errors"]) for g in reals], kw
        assert [int(v) for v in be] == [int(g["bit_errors"]) for g in reals], kw
    assert seen == {"blast", "mrc", "mrt", "alamouti"}
