"""GPU: config 4 staged through HBM (scripts/bench_staged_c4.py: one batched kernel per reference operator, draws of the
mcle-philox-v1 ledger) against the fused pipeline on the same realization indices and against the oracle chain."""
import os
import sys

import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
pytestmark = pytest.mark.gpu


def _staged(engine, first, count, dtype, method, nv, seed, labels=np.uint8):
    import bench_staged_c4 as st
    cnt = engine.new_counters()
    st.chain(engine, first, count, cnt, dtype, method, noise_var=nv, seed=seed, labels=labels)
    return engine.read_counters(cnt)


@pytest.mark.parametrize("method", [_lib.DEMOD_MINDIST, _lib.DEMOD_QAM_SLICER])
def test_staged_c4_counts_equal_the_fused_kernel_and_the_oracle(engine, method):
    import bench_staged_c4 as st
    st.bind(engine)
    seed, first, count = 424242, (1 << 33) + 5, 6
    nv = 1.0 / omodem.dB2Linear(25.0)
    kw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(seed, r), **kw) for r in range(first, first + count)]
    want_se = sum(w["symbol_errors"] for w in want)
    want_be = sum(w["bit_errors"] for w in want)
    got = _staged(engine, first, count, "f64", method, nv, seed)
    assert got == _staged(engine, first, count, "f64", method, nv, seed, labels=np.int32)     # byte labels == int32 labels
    assert got["n_realizations"] == count and got["n_symbols"] == 4096
    assert got["sym_errors"] == want_se and got["bit_errors"] == want_be
    assert got["sym_errors_sq"] == sum(w["symbol_errors"] ** 2 for w in want)
    fused = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, seed, first, count, method=method, dtype="f64")
    for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations"):
        assert got[k] == fused[k], k
    # complex64: the two paths round differently (the fused kernel folds the FFT scale into G); same statistics
    n = 2048
    a = _staged(engine, 0, n, "f32", method, nv, seed)
    b = engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, seed, 0, n, method=method, dtype="f32")
    assert a["n_realizations"] == b["n_realizations"] == n
    assert abs(a["sym_errors"] - b["sym_errors"]) <= 1e-5 * n * 4096 + 2


def test_batched_draw_operators_match_the_single_realization_ones(engine):
    """mcle_randn_c_batch / mcle_mimo_channel_philox against mcle_randn_c and mcle_mimo_channel on the same draws."""
    seed, first, count = 99, 1234567, 5
    for dt, tol in (("f64", 0.0), ("f32", 0.0)):
        Hb = engine.randn_c_batch(16, seed, first, count, stream=_lib.STREAM_CHAN, dtype=dt).get()
        for r in range(count):
            want = engine.randn_c(16, seed, first + r, stream=_lib.STREAM_CHAN, dtype=dt)
            assert np.array_equal(Hb[r], want)
        rs = np.random.RandomState(3)
        for nr, nt, ns in ((4, 4, 1040), (2, 3, 17), (4, 4, 33), (1, 2, 1)):
            cdt = np.complex128 if dt == "f64" else np.complex64
            H = (rs.randn(count, nr, nt) + 1j * rs.randn(count, nr, nt)).astype(cdt)
            X = (rs.randn(count, nt, ns) + 1j * rs.randn(count, nt, ns)).astype(cdt)
            got = engine.mimo_channel_philox(H, X, seed, first, 0.37, dtype=dt)
            noise = np.stack([engine.randn_c(nr * ns, seed, first + r, stream=_lib.STREAM_NOISE, dtype=dt).reshape(nr, ns)
                              for r in range(count)])
            want = engine.mimo_channel(H, X, noise=noise, noise_var=0.37, dtype=dt)
            scale = np.max(np.abs(want))
            assert np.max(np.abs(got - want)) <= (1e-13 if dt == "f64" else 2e-6) * scale, (dt, nr, nt, ns)
