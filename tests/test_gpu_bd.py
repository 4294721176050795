"""GPU: block diagonalisation, water-filling, pinv and the fused CoMP pipeline (SURVEY.md 8(f).3 tail)
against the reference's fixture (tests/golden/f6_block_diag.npz), the oracle (oracle/bd.py) and the
properties the reference's own tests check (tests/comm_package_test.py:47-332)."""
import numpy as np
import pytest

from helpers import golden_cases, relerr
from oracle import bd as obd, chains
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 20260927


def test_waterfilling_operator(engine):
    # the reference's known answer (tests/comm_package_test.py:47-85)
    gains = np.array([1.90, 1.76, 1.76, 1.35, 1.35, .733, .733, .100]) ** 2
    P, mu = engine.waterfilling(gains, 8.0, 0.181)
    assert abs(P.sum() - 8.0) < 1e-12 and abs(mu - 1.29134061296) < 1e-9
    np.testing.assert_array_almost_equal(P, [1.24120211, 1.23290828, 1.23290828, 1.19202648, 1.19202648,
                                             0.95446418, 0.95446418, 0.])
    rs = np.random.RandomState(4)
    for n in (1, 2, 3, 6, 8, 13, 64):
        g = rs.uniform(0.01, 5.0, size=(40, n))
        g[3] = g[3, 0]                                    # all equal
        for Pt, nv in ((1.0, 1.0), (0.05, 2.0), (30.0, 1e-3), (3.0, 1e-50)):
            P, mu = engine.waterfilling(g, Pt, nv)
            for b in range(g.shape[0]):
                want_P, want_mu = obd.waterfilling(g[b], Pt, nv)
                assert np.allclose(P[b], want_P, rtol=1e-12, atol=1e-15 * Pt), (n, Pt, nv, b)
                assert abs(mu[b] - want_mu) <= 1e-12 * abs(want_mu)
                assert np.array_equal(P[b] == 0, want_P == 0)
    with pytest.raises(ValueError):
        engine.waterfilling(np.ones(65), 1.0, 1.0)
    with pytest.raises(ValueError):
        engine.waterfilling(np.ones(4), 0.0, 1.0)


def test_pinv_operator(engine):
    rs = np.random.RandomState(9)
    for m, n in ((1, 1), (2, 2), (6, 6), (8, 8), (4, 6), (6, 3), (8, 2)):
        A = rs.randn(12, m, n) + 1j * rs.randn(12, m, n)
        A[1, :, 0] = 0                                     # a zero column -> a zero row
        if n > 1:
            A[2, :, 1] = A[2, :, 0] * (0.3 - 0.2j)         # rank deficient
        got = engine.pinv(A)
        for b in range(A.shape[0]):
            want = np.linalg.pinv(A[b])
            assert relerr(got[b], want) <= 1e-10 * max(1.0, np.max(np.abs(want))), (m, n, b)
        assert np.all(got[1][0] == 0)
    assert engine.pinv(A[0]).shape == (n, m)
    with pytest.raises(ValueError):
        engine.pinv(np.ones((9, 2), dtype=complex))


def test_block_diagonalize_matches_the_reference_fixture(engine):
    """Precoders of the reference (LAPACK's phases removed column by column), |newH|, the receive filter
    and the singular values, including water-filling that switches streams off."""
    dropped = 0
    for kw, reals in golden_cases("f6_block_diag"):
        K = kw["K"]
        H = np.stack([g["big_H"] for g in reals])
        out = engine.block_diagonalize(H, K, kw["iPu"], kw["bd_noise_var"], kw["waterfill"])
        assert not out["skipped"].any()
        for b, g in enumerate(reals):
            assert relerr(out["Ms"][b], obd.canonical_columns(g["Ms"])) <= 1e-9, kw
            assert relerr(np.abs(out["newH"][b]), np.abs(g["newH"])) <= 1e-9
            assert relerr(out["newH"][b], H[b] @ out["Ms"][b]) <= 1e-12
            _, Ms_c, W_c = obd.block_diagonalize_closed(H[b], K, kw["iPu"], kw["bd_noise_var"], kw["waterfill"])
            assert relerr(out["Ms"][b], Ms_c) <= 1e-10
            assert relerr(out["W"][b], W_c) <= 1e-9 * max(1.0, np.max(np.abs(W_c)))
            assert relerr(np.abs(out["W"][b]), np.abs(g["W"])) <= 1e-8 * max(1.0, np.max(np.abs(g["W"])))
            assert relerr(out["sigma"][b], obd.bd_no_power_scaling(H[b], K)[1]) <= 1e-10
            dropped += int(np.sum(np.sum(np.abs(out["Ms"][b]), axis=0) == 0))
    assert dropped > 0
    # a singular channel is reported, not solved
    Hs = np.ones((1, 4, 4), dtype=complex)
    assert engine.block_diagonalize(Hs, 2, 1.0, 0.1)["skipped"][0] == 1
    for bad in ((3, 2, np.zeros((1, 5, 5))), (5, 2, np.zeros((1, 10, 10))), (1, 5, np.zeros((1, 5, 5)))):
        with pytest.raises(ValueError):
            engine.block_diagonalize(bad[2].astype(complex), bad[0], 1.0, 0.1)


def test_block_diagonalize_random_sweep(engine):
    """Many random channels of every supported geometry, well and badly conditioned: finite output, exact
    block-diagonal structure, the power constraint, W newH = 1 on the active streams, and agreement with the
    oracle's SVD-free formulation to a tolerance that scales with the channel's condition number."""
    rs = np.random.RandomState(21)
    for K, r in ((2, 1), (8, 1), (2, 2), (3, 2), (4, 2), (2, 3), (2, 4)):
        n = K * r
        H = (rs.randn(300, n, n) + 1j * rs.randn(300, n, n)) / np.sqrt(2)
        H[:40] += 3.0 * (rs.randn(40, n, 1) + 1j * rs.randn(40, n, 1)) * np.ones((1, 1, n))   # nearly rank one
        for wf, nv in ((True, 0.3), (False, 0.0)):
            out = engine.block_diagonalize(H, K, 1.7, nv, wf)
            assert not out["skipped"].any()
            for key in ("Ms", "newH", "W", "sigma"):
                assert np.isfinite(out[key]).all(), (K, r, key)
            mask = np.kron(np.eye(K), np.ones((r, r))).astype(bool)
            for b in range(0, 300, 7):
                cond = np.linalg.cond(H[b])
                tol = 1e-12 * max(cond, 1.0) ** 2 + 1e-11
                Ms, newH, W = out["Ms"][b], out["newH"][b], out["W"][b]
                scale = np.max(np.abs(newH))
                assert np.max(np.abs(newH[~mask])) <= tol * scale, (K, r, b, cond)
                pw = [np.linalg.norm(Ms[:, u * r:(u + 1) * r]) ** 2 for u in range(K)]
                assert max(pw) <= 1.7 * (1 + 1e-9) and abs(max(pw) - 1.7) <= 1e-9 * 1.7
                active = np.sum(np.abs(Ms), axis=0) > 0
                assert relerr(np.diag(W @ newH)[active], np.ones(int(active.sum()))) <= tol
                assert np.all(W[~active] == 0)
                _, Ms_c, _ = obd.block_diagonalize_closed(H[b], K, 1.7, nv, wf)
                assert relerr(Ms, Ms_c) <= max(tol, 1e-9) * 10, (K, r, b, cond)


def test_mirror_classes_pass_the_reference_property_tests(engine):
    """tests/comm_package_test.py:112-332 on the mirror of pyphysim.comm.blockdiagonalization."""
    from pyphysim_amd.comm import blockdiagonalization as mbd, waterfilling as mwf
    rs = np.random.RandomState(12)
    Pu, noise_var, K, na = 5.0, 1e-6, 3, 2
    n = K * na
    BD = mbd.BlockDiagonalizer(K, Pu, noise_var, engine=engine)
    mask = np.kron(np.eye(K), np.ones((na, na))).astype(bool)

    def channel():
        return (rs.randn(n, n) + 1j * rs.randn(n, n)) / np.sqrt(2)

    def user_powers(Ms):
        return [np.linalg.norm(Ms[:, u * na:(u + 1) * na], "fro") ** 2 for u in range(K)]

    H = channel()
    Ms_bad, Sigma = BD._calc_BD_matrix_no_power_scaling(H)
    assert abs(np.linalg.norm(Ms_bad, "fro") ** 2 - n) < 1e-9
    assert np.abs((H @ Ms_bad)[~mask]).sum() < 1e-9
    assert relerr(Sigma, obd.bd_no_power_scaling(H, K)[1]) <= 1e-10
    Ms_g = BD._perform_global_waterfilling_power_scaling(Ms_bad, Sigma)
    assert Ms_g.shape == Ms_bad.shape and abs(np.linalg.norm(Ms_g, "fro") ** 2 - K * Pu) < 1e-9
    assert np.abs((H @ Ms_g)[~mask]).sum() < 1e-9
    Ms_n = BD._perform_normalized_waterfilling_power_scaling(Ms_bad, Sigma)
    assert np.linalg.norm(Ms_n, "fro") ** 2 <= K * Pu + 1e-9 and max(user_powers(Ms_n)) <= Pu + 1e-9
    for fn in (lambda h: mbd.block_diagonalize(h, K, Pu, noise_var, engine=engine), BD.block_diagonalize,
               BD.block_diagonalize_no_waterfilling):
        H = channel()
        newH, Ms = fn(H)
        assert np.abs(newH[~mask]).sum() < 1e-9
        assert np.linalg.norm(Ms, "fro") ** 2 <= K * Pu + 1e-9
        assert max(user_powers(Ms)) <= Pu + 1e-9
        assert abs(max(user_powers(Ms)) - Pu) < 1e-9          # the strongest block meets the budget
        W = mbd.calc_receive_filter(newH, engine=engine)
        assert relerr(W @ newH, np.eye(n)) <= 1e-9
        for u in range(K):
            blk = slice(u * na, (u + 1) * na)
            assert relerr(W[blk, blk] @ newH[blk, blk], np.eye(na)) <= 1e-9
    with pytest.raises(AssertionError):
        BD.block_diagonalize(np.ones((5, 5), dtype=complex))
    with pytest.raises(NotImplementedError):
        BD.block_diagonalize(np.ones((6, 9), dtype=complex))
    P, mu = mwf.doWF(np.array([1.0, 0.5, 0.25]), 3.0, 1.0, engine=engine)
    assert np.allclose(P, [2.0, 1.0, 0.0]) and abs(mu - 3.0) < 1e-12
    P2, mu2 = mwf.doWF(np.array([1.0, 0.5, 0.25]) / 2.0, 3.0, 1.0, Es=2.0, engine=engine)
    assert np.allclose(P2, P) and abs(mu2 - (P2[0] + 2.0)) < 1e-12


BD_CASES = [
    dict(mod="psk", M=4, K=3, nr=2, NSymbs=100, iPu=1.0, noise_var=0.03, bd_noise_var=1e-50, pathloss=None,
         waterfill=True),
    dict(mod="qam", M=16, K=3, nr=2, NSymbs=70, iPu=2.0, noise_var=0.01, bd_noise_var=0.5,
         pathloss=((1.0, 0.2, 0.05), (0.3, 1.0, 0.1), (0.02, 0.4, 1.0)), waterfill=True),
    dict(mod="qam", M=16, K=2, nr=2, NSymbs=64, iPu=1.0, noise_var=0.02, bd_noise_var=3.0, pathloss=None,
         waterfill=True),
    dict(mod="psk", M=8, K=4, nr=2, NSymbs=40, iPu=1.5, noise_var=0.01, bd_noise_var=1e-3, pathloss=None,
         waterfill=False),
    dict(mod="qam", M=16, K=2, nr=3, NSymbs=130, iPu=1.0, noise_var=0.01, bd_noise_var=0.1, pathloss=None,
         waterfill=True),
    dict(mod="qam", M=4, K=4, nr=1, NSymbs=33, iPu=1.0, noise_var=0.05, bd_noise_var=0.2, pathloss=None,
         waterfill=True),
    dict(mod="qam", M=64, K=2, nr=4, NSymbs=50, iPu=4.0, noise_var=0.002, bd_noise_var=0.05, pathloss=None,
         waterfill=True),
]


def _run(engine, kw, first, count, dtype, per_realization=True, method=_lib.DEMOD_MINDIST):
    engine.set_constellation(chains.constellation(kw["mod"], kw["M"]),
                             _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC)
    return engine.run_bd(kw["K"], kw["nr"], kw["NSymbs"], kw["iPu"], kw["noise_var"], SEED, first, count,
                         bd_noise_var=kw["bd_noise_var"], pathloss=kw["pathloss"], waterfilling=kw["waterfill"],
                         method=method, dtype=dtype, per_realization=per_realization)


@pytest.mark.parametrize("case", range(len(BD_CASES)))
def test_run_bd_against_the_oracle_chain(engine, case):
    """Common random numbers: the f64 pipeline reproduces the oracle chain's per-realization error counts (the
    chain in its canonical-phase form, oracle/chains.py chain_bd); f32 within the north-star tolerance; the
    reference's own formulation (LAPACK phases) gives the same statistics."""
    kw = BD_CASES[case]
    first, count = 5, 70                       # a full chunk of 64 plus a ragged one
    want = [chains.chain_bd(chains.PhiloxRng(SEED, r), canonical=True, **kw) for r in range(first, first + count)]
    want_se = np.array([w["symbol_errors"] for w in want])
    want_be = np.array([w["bit_errors"] for w in want])
    res, se, be = _run(engine, kw, first, count, "f64")
    n = kw["K"] * kw["nr"]
    assert res["n_realizations"] == count and res["n_skipped"] == 0
    assert res["n_symbols"] == n * kw["NSymbs"]
    assert np.array_equal(se, want_se) and np.array_equal(be, want_be), kw
    assert res["sym_errors"] == int(want_se.sum()) and res["bit_errors"] == int(want_be.sum())
    res32, se32, _ = _run(engine, kw, first, count, "f32")
    assert abs(int(se32.sum()) - int(want_se.sum())) / (count * n * kw["NSymbs"]) <= 1e-4
    # shards: any split of the realization range gives the same counters
    a = _run(engine, kw, first, 23, "f64", per_realization=False)
    b = _run(engine, kw, first + 23, count - 23, "f64", per_realization=False)
    assert a["sym_errors"] + b["sym_errors"] == res["sym_errors"]
    assert a["bit_errors_sq"] + b["bit_errors_sq"] == res["bit_errors_sq"]
    # statistics of the reference's formulation
    ref = [chains.chain_bd(chains.PhiloxRng(SEED, r), canonical=False, **kw)["symbol_errors"]
           for r in range(first, first + count)]
    ser_ref, ser = np.sum(ref) / (count * n * kw["NSymbs"]), want_se.sum() / (count * n * kw["NSymbs"])
    assert abs(ser - ser_ref) <= 0.35 * ser_ref + 5e-3


def test_run_bd_noiseless_and_errors(engine):
    kw = dict(BD_CASES[0], noise_var=0.0)
    assert _run(engine, kw, 0, 300, "f32", per_realization=False)["sym_errors"] == 0
    assert _run(engine, kw, 0, 300, "f64", per_realization=False)["sym_errors"] == 0
    assert _run(engine, dict(BD_CASES[3], noise_var=0.0), 0, 100, "f32", per_realization=False)["sym_errors"] == 0
    for bad in (dict(K=5, nr=2), dict(K=1, nr=5), dict(K=3, nr=3), dict(NSymbs=0), dict(iPu=0.0)):
        with pytest.raises(Exception):
            _run(engine, dict(BD_CASES[0], **bad), 0, 4, "f32")
    with pytest.raises(ValueError):
        engine.run_bd(3, 2, 10, 1.0, 0.1, SEED, 0, 4, pathloss=np.ones((2, 2)))
    # an empty range is a no-op on every chunked pipeline
    z = _run(engine, BD_CASES[0], 0, 0, "f32", per_realization=False)
    assert z["n_realizations"] == 0 and z["sym_errors"] == 0
    assert engine.run_ia(200, 0.1, SEED, 7, 0)["n_realizations"] == 0
    assert engine.run_mimo_flat("blast", 2, 2, 64, 0.1, SEED, 7, 0)["n_realizations"] == 0
    assert engine.block_diagonalize(np.zeros((0, 4, 4), dtype=complex), 2, 1.0, 0.1)["Ms"].shape == (0, 4, 4)
    assert engine.waterfilling(np.zeros((0, 3)), 1.0, 1.0)[0].shape == (0, 3)


def test_bd_simulator_runs_the_comp_application(engine):
    """BdSimulator = apps/comp_BD/simulate_comp_simple.py as a BatchedSimulationRunner: SER falls with SNR and
    agrees with the oracle chain at one point."""
    from pyphysim_amd.simulators import BdSimulator
    sim = BdSimulator(SNR=[5.0, 15.0], modulator="psk", M=4, K=3, Nr=2, NSymbs=100, noise_var=1e-3,
                      path_loss_border=1.0, rep_max=2000, seed=3, dtype="f32", engine=engine)
    sim.simulate()
    ser = sim.results.get_result_values_list("ser")
    assert ser[0] > ser[1] > 0
    iPu = 10 ** 1.5 * 1e-3
    want = [chains.chain_bd(chains.PhiloxRng(1, r), mod="psk", M=4, K=3, nr=2, NSymbs=100, iPu=iPu, noise_var=1e-3)
            ["symbol_errors"] for r in range(300)]
    ser_ref = np.sum(want) / (300.0 * 600)
    assert abs(ser[1] - ser_ref) <= 0.3 * ser_ref + 2e-3


@pytest.mark.parametrize("K,nr", [(3, 2), (2, 2), (2, 3), (4, 1), (6, 1)])
def test_static_solve_equals_the_runtime_sized_solve(engine, K, nr):
    """csrc/bd_static.hpp (compile-time sizes, every matrix in registers; K nr <= 6) against the run-time-sized bd_solve it
    stands in for inside the fused pipeline (engine option bd_runtime_solve): per-realization counts equal in complex128 and
    complex64, water-filling on and off, with path loss."""
    from oracle import chains
    engine.set_constellation(chains.constellation("psk", 4), _lib.CONST_GENERIC)
    rs = np.random.RandomState(K * 10 + nr)
    pl = rs.uniform(0.2, 1.0, (K, K))
    for wf in (True, False):
        for pathloss in ((None, pl) if K <= 4 else (None,)):
            for dt in ("f64", "f32"):
                kw = dict(bd_noise_var=1e-3 if wf else 1e-50, pathloss=pathloss, waterfilling=wf, dtype=dt, per_realization=True)
                new = engine.run_bd(K, nr, 64, 1.0, 0.05, 555, 7, 700, **kw)
                with engine.options(bd_runtime_solve=1):
                    old = engine.run_bd(K, nr, 64, 1.0, 0.05, 555, 7, 700, **kw)
                assert new[0]["n_realizations"] == old[0]["n_realizations"] and new[0]["n_skipped"] == old[0]["n_skipped"]
                d = np.abs(new[1].astype(np.int64) - old[1].astype(np.int64))
                assert np.count_nonzero(d) <= 2 and d.max() <= 1, (K, nr, wf, dt, int(d.sum()))
