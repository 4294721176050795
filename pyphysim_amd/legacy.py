"""Same-seed parity mode: the reference's template simulators replayed on the GPU with the
reference's OWN random numbers.

``np.random.seed(s)`` in the reference fixes every draw of a realization (legacy MT19937,
SURVEY.md App. A.1).  :func:`Engine.legacy_draws` regenerates those draws on the device
(csrc/kernels_legacy.hip) and the chains below push them through the per-operator HIP kernels in
the f64 instantiation, in the reference's operator and draw order.  The error counts therefore
equal the reference's for the same seed -- not statistically, exactly -- which is what
tests/test_gpu_legacy.py checks against the reference-minted fixtures in tests/golden/.

This mode exists for validation (it is operator-granular and HBM-staged); the throughput path is
the fused Philox pipelines.
"""
import math

import numpy as np

from . import _lib
from .channels import discretize_profile
from .modulators import constellation, dB2Linear, level2bits

INV_SQRT2 = 1.0 / math.sqrt(2.0)


def _table(mod, M):
    kind = {"qam": _lib.CONST_QAM, "bpsk": _lib.CONST_BPSK}.get(mod, _lib.CONST_GENERIC)
    return constellation(mod, M) if mod in ("qam", "psk") else constellation(mod), kind


def run_awgn(eng, seed_base, first, count, mod="qam", M=16, N=10000, snr_db=10.0):
    """apps/awgn_modulators/simulate_psk.py:51-115 under np.random.seed(seed_base + r):
    randint(0, M, N); randn_c(N) = randn(N) real block, randn(N) imag block."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    noise_var = 1.0 / float(dB2Linear(snr_db))
    ints, dbls = eng.legacy_draws([("randint", N, M), ("randn", N), ("randn", N)], seed_base, first, count)
    h = dbls.get()
    noise = eng.complex_from_parts(h[:, :N].copy(), h[:, N:].copy(), INV_SQRT2, dtype="f64")
    tx = eng.modulate(ints, dtype="f64")
    rx = eng.awgn_add(tx, noise, noise_var, dtype="f64")
    cnt, se, be = eng.demod_count(rx, ints, n_real=count, dtype="f64")
    return cnt, se, be


def run_flat_jakes(eng, seed_base, first, count, mod="qam", M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8):
    """Config 2 template (SuChannel(JakesSampleGenerator)): randint(N); Jakes ctor rand(L,1) x2
    (one discarded sample); TdlChannel ctor rand(L,1,1) x2; randn_c(N)."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    noise_var = 1.0 / float(dB2Linear(snr_db))
    prog = [("randint", N, M), ("rand", 2 * L), ("rand", 2 * L), ("randn", N), ("randn", N)]
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    h_d = dbls.get()
    step = Ts * 1.0000000001
    dt = (Ts + step) - Ts                      # numpy.arange's effective increment (fading_generators.py:459-462)
    se, be = [], []
    for r in range(count):
        row = h_d[r]
        phi = 2 * np.pi * row[2 * L:3 * L].reshape(L, 1)
        psi = 2 * np.pi * row[3 * L:4 * L].reshape(L, 1)
        hh = eng.jakes_generate(phi, psi, Fd, Ts, dt, N, dtype="f64", device=True)
        noise = eng.complex_from_parts(row[4 * L:4 * L + N].copy(), row[4 * L + N:4 * L + 2 * N].copy(), INV_SQRT2,
                                       dtype="f64")
        idx = eng.to_device(ints.get()[r], np.int32)
        tx = eng.modulate(idx, dtype="f64")
        faded = eng.tdl_apply(tx, hh.reshape(N), [0], dtype="f64")
        rx = eng.awgn_add(faded, noise, noise_var, dtype="f64")
        eq = eng.cdiv(rx, hh.reshape(N), dtype="f64")
        _, s, b = eng.demod_count(eq, idx, n_real=1, dtype="f64")
        se.append(int(s[0]))
        be.append(int(b[0]))
    return np.array(se), np.array(be)


def run_flat_rayleigh(eng, seed_base, first, count, mod="qam", M=16, N=1000, snr_db=15.0, form="suchannel"):
    """Flat i.i.d. Rayleigh fading under np.random.seed(seed_base + r), in either of the
    reference's two statements of it:
      'suchannel'  SuChannel(RayleighSampleGenerator()): randint(N); the generator's ctor randn_c() (one discarded
                   sample = two scalar randn calls, i.e. ONE polar pair); corrupt_data randn_c(1, N) = h; randn_c(N)
                   noise (fading_generators.py:226-249, singleuser.py:48-81)
      'notebook'   notebooks/Transmission_with_Rayleigh_and_AWGN_channels.ipynb cell 8: randint(N); noise
                   randn_c(N); h = randn_c(N)."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    noise_var = 1.0 / float(dB2Linear(snr_db))
    if form == "notebook":
        prog = [("randint", N, M)] + [("randn", N)] * 4
        o_noise, o_h = 0, 2 * N
    else:
        prog = [("randint", N, M), ("randn", 1), ("randn", 1)] + [("randn", N)] * 4
        o_h, o_noise = 2, 2 + 2 * N
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    d = dbls.get()
    h = eng.complex_from_parts(d[:, o_h:o_h + N].copy(), d[:, o_h + N:o_h + 2 * N].copy(), INV_SQRT2, dtype="f64")
    noise = eng.complex_from_parts(d[:, o_noise:o_noise + N].copy(), d[:, o_noise + N:o_noise + 2 * N].copy(),
                                   INV_SQRT2, dtype="f64")
    tx = eng.modulate(ints, dtype="f64")
    rx = eng.awgn_add(eng.cmul(h, tx, dtype="f64"), noise, noise_var, dtype="f64")
    eq = eng.cdiv(rx, h, dtype="f64")
    cnt, se, be = eng.demod_count(eq, ints, n_real=count, dtype="f64")
    return cnt, se, be


def run_ofdm_tdl(eng, seed_base, first, count, mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None,
                 n_ofdm_sym=1, snr_db=20.0, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8,
                 tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays_samples=(0, 1, 2, 3, 4)):
    """Config 3 template (notebooks/TDL_and_OFDM.ipynb): randint; Jakes ctor rand x2; TdlChannel ctor
    rand(L, taps, 1) x2; randn_c over the faded length (n + max delay)."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    used = num_used or fft_size
    noise_var = 1.0 / float(dB2Linear(snr_db))
    p_lin, d_idx = discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                      np.asarray(tap_delays_samples, dtype=float) * Ts, Ts)
    S = len(d_idx)
    n_data = used * n_ofdm_sym
    n = n_ofdm_sym * (fft_size + cp_size)
    n_noise = n + int(d_idx[-1])
    prog = [("randint", n_data, M), ("rand", 2 * L), ("rand", 2 * L * S), ("randn", n_noise), ("randn", n_noise)]
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    h_d, h_i = dbls.get(), ints.get()
    step = Ts * 1.0000000001
    dt = (Ts + step) - Ts
    se, be = [], []
    for r in range(count):
        row = h_d[r]
        o = 2 * L
        phi = 2 * np.pi * row[o:o + L * S].reshape(L, S)
        psi = 2 * np.pi * row[o + L * S:o + 2 * L * S].reshape(L, S)
        o += 2 * L * S
        noise = eng.complex_from_parts(row[o:o + n_noise].copy(), row[o + n_noise:o + 2 * n_noise].copy(), INV_SQRT2,
                                       dtype="f64")
        idx = eng.to_device(h_i[r], np.int32)
        sym = eng.modulate(idx, dtype="f64")
        tx = eng.ofdm_modulate(sym, fft_size, cp_size, used, dtype="f64")
        taps = eng.jakes_generate(phi, psi, Fd, Ts, dt, n, tap_power=p_lin, dtype="f64", device=True)
        faded = eng.tdl_apply(tx.reshape(n), taps.reshape(S * n), d_idx, dtype="f64")
        rx = eng.awgn_add(faded, noise, noise_var, dtype="f64")
        rx_n = eng.to_device(rx.get()[:n], np.complex128)
        demod = eng.ofdm_demodulate(rx_n, fft_size, cp_size, used, dtype="f64")
        eq = eng.onetap_equalize(demod.reshape(n_data), taps.reshape(S * n), d_idx, fft_size, cp_size, used,
                                 dtype="f64")
        _, s, b = eng.demod_count(eq, idx, n_real=1, dtype="f64")
        se.append(int(s[0]))
        be.append(int(b[0]))
    return np.array(se), np.array(be)


def run_mimo_ofdm(eng, seed_base, first, count, mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16,
                  num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True):
    """Config 4 template (apps/mimo/simulate_mimo.py:68-142 + per-antenna OFDM): H = randn_c(nr, nt);
    randint(nt * used * n_sym); randn_c(nr, n_sym * (fft + cp)).  Batched over realizations."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    used = num_used or fft_size
    noise_var = 1.0 / float(dB2Linear(snr_db))
    n_data = nt * used * n_ofdm_sym
    ns = n_ofdm_sym * (fft_size + cp_size)
    prog = [("randn", nr * nt), ("randn", nr * nt), ("randint", n_data, M), ("randn", nr * ns), ("randn", nr * ns)]
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    h = dbls.get()
    o = 2 * nr * nt
    H = eng.complex_from_parts(h[:, :nr * nt].copy(), h[:, nr * nt:o].copy(), INV_SQRT2, dtype="f64")
    noise = eng.complex_from_parts(h[:, o:o + nr * ns].copy(), h[:, o + nr * ns:o + 2 * nr * ns].copy(), INV_SQRT2,
                                   dtype="f64")
    sym = eng.modulate(ints, dtype="f64")                                  # [count, n_data]
    X = eng.blast_encode(sym, nt, batch=count, dtype="f64")                # [count, nt, n_data/nt]
    T = eng.ofdm_modulate(X, fft_size, cp_size, used, batch=count * nt, dtype="f64")
    R = eng.mimo_channel(H.reshape(count, nr, nt), T.reshape(count, nt, ns), noise.reshape(count, nr, ns),
                         noise_var, dtype="f64")
    Y = eng.ofdm_demodulate(R, fft_size, cp_size, used, batch=count * nr, dtype="f64")
    G, skipped = eng.blast_filter(H.reshape(count, nr, nt), noise_var if mmse else 0.0, dtype="f64")
    est = eng.blast_decode(G, Y.reshape(count, nr, used * n_ofdm_sym), dtype="f64")
    cnt, se, be = eng.demod_count(est, ints, n_real=count, dtype="f64")
    return cnt, se, be


def run_ia(eng, seed_base, first, count, mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0,
           algo="closed_form", max_iterations=50, relative_factor=1e-6, initialize_with="random"):
    """Config 5 / SURVEY 8(f).3 template (apps/ia/simulate_ia.py:94-245).  The reference draws from separate
    RandomStates -- the channel's (multiuser.py:670-709: randn_c(6, 6)), the noise's (randn_c(6, NSymbs)), the
    iterative solvers' own (iabase.py:95,538-540: randn_c(2, 1) per user) and the global one (randint(0, M,
    [3, NSymbs])) -- all seeded with seed_base + r by the harness that minted the fixtures."""
    if (K, nr, nt, Ns) != (3, 2, 2, 1):
        raise ValueError("the IA kernels cover K = 3, 2x2, one stream per user")
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    noise_var = 1.0 / float(dB2Linear(snr_db))
    _, ch = eng.legacy_draws([("randn", 36), ("randn", 36)], seed_base, first, count)
    _, nz = eng.legacy_draws([("randn", 6 * NSymbs), ("randn", 6 * NSymbs)], seed_base, first, count)
    ints, _ = eng.legacy_draws([("randint", 3 * NSymbs, M)], seed_base, first, count)
    hc = ch.get()
    big_H = INV_SQRT2 * (hc[:, :36] + 1j * hc[:, 36:]).reshape(count, 6, 6)
    if algo == "closed_form":
        sol = eng.ia_closed_form(big_H, noise_var)
    else:
        _, fi = eng.legacy_draws([("randn", 2), ("randn", 2)] * 3, seed_base, first, count)
        hf = fi.get().reshape(count, 3, 2, 2)                    # [r, user, re/im block, antenna]
        F0 = INV_SQRT2 * (hf[:, :, 0, :] + 1j * hf[:, :, 1, :])
        F0 = F0 / np.linalg.norm(F0, axis=2, keepdims=True)
        sol = eng.ia_iterative(algo, big_H, F0, noise_var, max_iterations, relative_factor,
                               "fix" if initialize_with == "random" else initialize_with)
    # precoding and receive filtering as block matrices through the multi-user channel kernel
    Fbig = np.zeros((count, 6, 3), dtype=complex)
    Ubig = np.zeros((count, 3, 6), dtype=complex)
    for k in range(3):
        Fbig[:, 2 * k:2 * k + 2, k] = sol["F"][:, k]
        Ubig[:, k, 2 * k:2 * k + 2] = sol["U"][:, k]
    hn = nz.get()
    n6 = 6 * NSymbs
    noise = eng.complex_from_parts(hn[:, :n6].copy(), hn[:, n6:].copy(), INV_SQRT2, dtype="f64").reshape(count, 6, NSymbs)
    sym = eng.modulate(ints, dtype="f64").reshape(count, 3, NSymbs)
    X = eng.mimo_channel(eng.to_device(Fbig), sym, dtype="f64")
    Y = eng.mimo_channel(eng.to_device(big_H), X, noise, noise_var, dtype="f64")
    est = eng.mimo_channel(eng.to_device(Ubig), Y, dtype="f64")
    cnt, se, be = eng.demod_count(est.reshape(count, 3 * NSymbs), ints, n_real=count, dtype="f64")
    return cnt, sol, se, be


def run_mimo_ofdm_tdl(eng, seed_base, first, count, mod="qam", M=16, nt=2, nr=2, fft_size=64, cp_size=16, num_used=None,
                      n_ofdm_sym=2, snr_db=20.0, Fd=50.0, Ts=1e-6, L=8, tap_powers_dB=(0.0, -4.0, -9.0),
                      tap_delays_samples=(0, 2, 5)):
    """SURVEY 8(f).1 template: randint(nt * used * n_sym); Jakes ctor with shape (nr, nt): rand(L, nr, nt, 1) x2
    (one discarded sample); TdlMimoChannel ctor: rand(L, taps, nr, nt, 1) x2; randn_c(nr, n + max delay)."""
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    used = num_used or fft_size
    noise_var = 1.0 / float(dB2Linear(snr_db))
    p_lin, d_idx = discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                      np.asarray(tap_delays_samples, dtype=float) * Ts, Ts)
    S = len(d_idx)
    P = S * nr * nt
    ns = used * n_ofdm_sym
    n = n_ofdm_sym * (fft_size + cp_size)
    n_out = n + int(d_idx[-1])
    prog = [("randint", nt * ns, M), ("rand", 2 * L * nr * nt), ("rand", 2 * L * P), ("randn", nr * n_out),
            ("randn", nr * n_out)]
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    h_d = dbls.get()
    step = Ts * 1.0000000001
    dt = (Ts + step) - Ts
    sym = eng.modulate(ints, dtype="f64")
    X = eng.blast_encode(sym, nt, batch=count, dtype="f64")
    T = eng.ofdm_modulate(X, fft_size, cp_size, used, batch=count * nt, dtype="f64").reshape(count, nt, n)
    T_h = T.get()
    se, be = [], []
    for r in range(count):
        row = h_d[r]
        o = 2 * L * nr * nt
        phi = 2 * np.pi * row[o:o + L * P].reshape(L, P)
        psi = 2 * np.pi * row[o + L * P:o + 2 * L * P].reshape(L, P)
        o += 2 * L * P
        noise = eng.complex_from_parts(row[o:o + nr * n_out].copy(), row[o + nr * n_out:o + 2 * nr * n_out].copy(),
                                       INV_SQRT2, dtype="f64").reshape(nr, n_out)
        taps = eng.jakes_generate(phi, psi, Fd, Ts, dt, n, tap_power=np.repeat(p_lin, nr * nt), dtype="f64",
                                  device=True).reshape(S, nr, nt, n)
        faded = eng.tdl_apply_mimo(eng.to_device(T_h[r]), taps, d_idx, dtype="f64")
        R = eng.awgn_add(faded, noise, noise_var, dtype="f64")
        Rn = eng.slice_rows(R, n) if n_out != n else R
        Y = eng.ofdm_demodulate(Rn, fft_size, cp_size, used, batch=nr, dtype="f64").reshape(nr, ns)
        Hu = eng.tdl_mean_freq_response(taps, d_idx, n_ofdm_sym, fft_size, cp_size, used, dtype="f64")
        G, _ = eng.blast_filter(Hu.reshape(ns, nr, nt), noise_var, dtype="f64")
        est = eng.blast_decode_per_subcarrier(G, Y, dtype="f64")
        idx_r = eng.to_device(ints.get()[r], np.int32)
        _, s, b = eng.demod_count(est, idx_r, n_real=1, dtype="f64")
        se.append(int(s[0]))
        be.append(int(b[0]))
    return np.array(se), np.array(be)


def run_mimo_scheme(eng, seed_base, first, count, scheme="blast", mod="qam", M=16, nt=2, nr=2, NSymbs=200, snr_db=15.0):
    """apps/mimo/simulate_mimo.py:68-100 under np.random.seed(seed_base + r): randn_c(Nr, Nt); randint(0, M,
    NSymbs * layers); randn_c(Nr, NSymbs).  Encode / decode through the mirror classes (per-operator kernels, f64).
    Blast, MRC, MRT and Alamouti reproduce the reference's counts; SVD / GMD only statistically (their
    singular-vector phases are LAPACK's choice there, the Jacobi routine's here)."""
    from . import mimo as mm
    table, kind = _table(mod, M)
    eng.set_constellation(table, kind)
    noise_var = 1.0 / float(dB2Linear(snr_db))
    cls = {"blast": mm.Blast, "mrc": mm.MRC, "mrt": mm.MRT, "alamouti": mm.Alamouti, "svd": mm.SVDMimo,
           "gmd": mm.GMDMimo}[scheme]
    layers = 1 if scheme in ("alamouti", "mrt") else nt
    prog = [("randn", nr * nt), ("randn", nr * nt), ("randint", NSymbs * layers, M), ("randn", nr * NSymbs),
            ("randn", nr * NSymbs)]
    ints, dbls = eng.legacy_draws(prog, seed_base, first, count)
    h, idx = dbls.get(), ints.get()
    se, be = [], []
    o = 2 * nr * nt
    for r in range(count):
        H = INV_SQRT2 * (h[r, :nr * nt] + 1j * h[r, nr * nt:o]).reshape(nr, nt)
        noise = INV_SQRT2 * (h[r, o:o + nr * NSymbs] + 1j * h[r, o + nr * NSymbs:o + 2 * nr * NSymbs]).reshape(nr, NSymbs)
        obj = cls(H, engine=eng, dtype="f64")
        sym = eng.modulate(idx[r], dtype="f64")
        X = np.asarray(obj.encode(sym))
        Y = eng.mimo_channel(H[np.newaxis], X.reshape(1, nt, -1), noise[np.newaxis], noise_var, dtype="f64")[0]
        est = np.asarray(obj.decode(Y)).reshape(-1)
        _, s, b = eng.demod_count(est, idx[r], n_real=1, dtype="f64")
        se.append(int(s[0]))
        be.append(int(b[0]))
    return np.array(se), np.array(be)
