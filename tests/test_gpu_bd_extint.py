"""GPU: block diagonalisation with external interference (csrc/kernels_bd.hip k_bd_extint) against the reference's own
WhiteningBD / EnhancedBD runs (tests/golden/f6b_bd_extint.npz, minted by oracle/make_golden.py from
comm/blockdiagonalization.py:666-1469 on seeded MultiUserChannelMatrixExtInt channels).

Singular / eigen-vector phases are the kernel's, so solutions are compared through what is phase free: the stream counts,
Ms Ms^H, W^H W and the modulus of the equivalent channel W H_k Ms (identity-like for the inverting filters).

EnhancedBD's stream reduction (metrics 'fixed', 'capacity', 'effective_throughput') is a special case: MsPk = Ms_k Pk
mixes the COLUMNS of the BD directions Ms_k with weights taken from the interference covariance, so the result depends
on the phase each column of Ms_k happened to get from the SVD -- LAPACK's in the reference, the kernel's here; neither is
canonical.  Those variants are pinned against the oracle's restatement of the reference arithmetic (itself asserted equal
to the reference when the fixture is minted) run on the kernel's own Ms_k, plus the structural properties."""
import numpy as np
import pytest

from helpers import GOLDEN, relerr

pytestmark = pytest.mark.gpu


def _cases():
    z = np.load(GOLDEN + "/f6b_bd_extint.npz", allow_pickle=False)
    out = []
    for ci in range(int(z["n_cases"])):
        vi = 0
        while "case%d_v%d_name" % (ci, vi) in z.files:
            out.append((ci, vi))
            vi += 1
    return z, out


Z, CASES = _cases()


@pytest.mark.parametrize("ci,vi", CASES)
def test_bd_with_external_interference_against_reference(engine, ci, vi):
    from pyphysim_amd import modulators, multiuser
    from pyphysim_amd.comm import blockdiagonalization as bd
    K, r, n_ext = [int(v) for v in Z["case%d_cfg" % ci]]
    iPu, nv, pe = [float(v) for v in Z["case%d_par" % ci]]
    method, metric, ns = str(Z["case%d_v%d_name" % (ci, vi)]).split("/")
    muc = multiuser.MultiUserChannelMatrixExtInt(engine=engine)
    muc.init_from_channel_matrix(Z["case%d_big_H" % ci], r, r, K, n_ext)
    muc.noise_var = nv
    assert muc.big_H_no_ext_int.shape == (K * r, K * r) and muc.K == K
    if method == "whitening":
        Ms, W, Ns = bd.WhiteningBD(K, iPu, nv, pe, engine=engine).block_diagonalize_no_waterfilling(muc)
    else:
        e = bd.EnhancedBD(K, iPu, nv, pe, engine=engine)
        extra = {"num_streams": int(ns)} if metric in ("naive", "fixed") else (
            {"modulator": modulators.QAM(16, engine=engine), "packet_length": 120}
            if metric == "effective_throughput" else None)
        e.set_ext_int_handling_metric(None if metric == "None" else metric, extra)
        Ms, W, Ns = e.block_diagonalize_no_waterfilling(muc)
    tag = "case%d_v%d_" % (ci, vi)
    reduced = method == "enhanced" and metric in ("fixed", "capacity", "effective_throughput")
    if reduced:
        from oracle import bd as obd
        from pyphysim_amd.modulators import QAM
        big_H = np.asarray(muc.big_H)
        Ms_bad = engine.block_diagonalize(big_H[:, :K * r], K, float(r), nv, False)["Ms"][0]     # unit-norm columns
        mod = QAM(16, engine=engine)
        mf = (lambda s: float(np.sum(mod.calcTheoreticalSpectralEfficiency(10.0 * np.log10(s), 120)))) \
            if metric == "effective_throughput" else None
        want = obd.enhanced_bd(big_H, K, r, r, iPu, nv, pe, metric, int(ns) if ns != "None" else None, mf, Ms_bad=Ms_bad)
        assert [int(n) for n in Ns] == [int(n) for n in want[2]]
        if metric == "fixed":
            assert [int(n) for n in Ns] == [int(ns)] * K == [int(n) for n in Z[tag + "Ns"]]
    else:
        assert [int(n) for n in Ns] == [int(n) for n in Z[tag + "Ns"]]
    for k in range(K):
        Hk = np.asarray(muc.big_H)[k * r:(k + 1) * r, :K * r]
        assert Ms[k].shape == (K * r, int(Ns[k])) and W[k].shape == (int(Ns[k]), r)
        if reduced:
            pm, pw, eq = want[0][k] @ want[0][k].conj().T, want[1][k].conj().T @ want[1][k], np.abs(want[1][k] @ Hk @ want[0][k])
            assert relerr(W[k] @ Hk @ Ms[k], np.eye(int(Ns[k]))) <= 1e-8          # the filter inverts the kept streams
        else:
            pm, pw, eq = Z[tag + "u%d_PM" % k], Z[tag + "u%d_PW" % k], Z[tag + "u%d_EQ" % k]
        assert relerr(Ms[k] @ Ms[k].conj().T, pm) <= 1e-8
        assert relerr(W[k].conj().T @ W[k], pw) <= 1e-8 * max(1.0, float(np.abs(pw).max()))
        assert relerr(np.abs(W[k] @ Hk @ Ms[k]), eq) <= 1e-7
        assert abs(np.linalg.norm(Ms[k], "fro") ** 2 - iPu) <= 1e-9 * iPu        # every user transmits iPu
        # block diagonalisation: user k's precoder does not reach the other users
        for l in range(K):
            if l != k:
                Hl = np.asarray(muc.big_H)[l * r:(l + 1) * r, :K * r]
                assert np.abs(Hl @ Ms[k]).max() <= 1e-9


def test_bd_extint_mirror_errors_and_batch(engine):
    from pyphysim_amd import multiuser
    from pyphysim_amd.comm import blockdiagonalization as bd
    e = bd.EnhancedBD(3, 1.0, 0.01, 1.0, engine=engine)
    with pytest.raises(AttributeError, match="num_streams"):
        e.set_ext_int_handling_metric("naive")
    with pytest.raises(AttributeError, match="modulator"):
        e.set_ext_int_handling_metric("effective_throughput", {"packet_length": 100})
    with pytest.raises(AttributeError):
        e.set_ext_int_handling_metric("nonsense")
    e.set_ext_int_handling_metric("capacity")
    assert e.metric_name == "capacity"
    muc = multiuser.MultiUserChannelMatrixExtInt(engine=engine)
    muc.set_channel_seed(3)
    muc.randomize(2, 2, 3, 2)
    muc.noise_var = 0.01
    R = muc.calc_cov_matrix_extint_plus_noise(0.5)
    Wf = e.calc_whitening_matrices(muc) if False else bd.WhiteningBD(3, 1.0, 0.01, 0.5, engine=engine).calc_whitening_matrices(muc)
    for k in range(3):
        assert relerr(Wf[k] @ R[k] @ Wf[k].conj().T, np.eye(2)) <= 1e-10          # it whitens
    # a batch through the engine equals one channel at a time
    H = np.stack([np.asarray(muc.big_H), 2.0 * np.asarray(muc.big_H)])
    both = engine.bd_extint(H, 3, 2, 2, 1.0, 0.01, 0.5, "enhanced", "capacity")
    one = engine.bd_extint(H[1], 3, 2, 2, 1.0, 0.01, 0.5, "enhanced", "capacity")
    assert np.array_equal(both["Ns"][1], one["Ns"][0]) and relerr(both["Ms"][1], one["Ms"][0]) <= 1e-12


def test_calc_receive_filter_user_k(engine):
    from oracle import bd as obd
    from pyphysim_amd.comm import blockdiagonalization as bd
    rs = np.random.RandomState(8)
    Heq = rs.randn(3, 2) + 1j * rs.randn(3, 2)
    P = np.linalg.qr(rs.randn(3, 2) + 1j * rs.randn(3, 2))[0]
    assert relerr(bd.EnhancedBD.calc_receive_filter_user_k(Heq, engine=engine), obd.ebd_receive_filter(Heq)) <= 1e-11
    assert relerr(bd.EnhancedBD.calc_receive_filter_user_k(Heq, P, engine=engine), obd.ebd_receive_filter(Heq, P)) <= 1e-10
