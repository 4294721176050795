"""GPU: seeded random configurations of every fused pipeline against the oracle chains under common random
numbers (mcle-philox-v1).  The f64 instantiation must reproduce the oracle's per-realization symbol and bit
error counts exactly, whatever the modulation, the sizes, the SNR or the position of the realization range;
the f32 instantiation stays within the north-star tolerance on the same draws."""
import numpy as np
import pytest

from oracle import chains, channels as och, modem as omodem
from pyphysim_amd import _lib

import os

pytestmark = pytest.mark.gpu
SEED = 20260927
N_TRIALS = int(os.environ.get("MCLE_FUZZ_TRIALS_BASE", "6"))        # raise for a long hunt on the GPU box
OFFSET = int(os.environ.get("MCLE_FUZZ_OFFSET", "0"))               # shifts every trial's random configuration
N_MFMA_TRIALS_EARLY = int(os.environ.get("MCLE_FUZZ_TRIALS", "10"))
MODS = [("bpsk", 2), ("qpsk", 4), ("psk", 8), ("psk", 16), ("qam", 4), ("qam", 16), ("qam", 64), ("qam", 256)]


def _bind(engine, mod, M):
    engine.set_constellation(chains.constellation(mod, M), _lib.CONST_QAM if mod == "qam" else
                             (_lib.CONST_BPSK if mod == "bpsk" else _lib.CONST_GENERIC))


def _oracle(fn, first, count, **kw):
    out = [fn(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + count)]
    return (np.array([o["symbol_errors"] for o in out]), np.array([o["bit_errors"] for o in out]),
            out[0]["num_symbols"], out[0]["num_bits"])


def _check(res, se, be, want, dt, what, iterative=False):
    want_se, want_be, nsym, nbits = want
    n = len(want_se)
    assert res["n_realizations"] == n and res["n_symbols"] == nsym and res["n_bits"] == nbits, what
    if dt == "f64" and iterative:
        # An iterative solver amplifies the rounding-level differences between its device eigen-solver (Jacobi) and LAPACK over
        # the iterations; where it ends badly conditioned (a stream in outage: an eighth or more of the realization's symbols
        # wrong, decisions at near-ties) a few decisions may differ -- found by a seed hunt (MCLE_FUZZ_OFFSET=7000, trial 148:
        # 119 against 117 errors of 387 in one 8-PSK realization after 21 iterations, the other eleven equal).  Every other
        # realization must be exact.
        outage = want_se >= nsym // 8
        assert np.array_equal(se[~outage], want_se[~outage]) and np.array_equal(be[~outage], want_be[~outage]), what
        assert np.all(np.abs(se[outage].astype(np.int64) - want_se[outage]) <= np.maximum(2, 0.02 * want_se[outage])), what
        assert np.all(np.abs(be[outage].astype(np.int64) - want_be[outage]) <= np.maximum(8, 0.04 * want_be[outage])), what
    elif dt == "f64":
        assert np.array_equal(se, want_se) and np.array_equal(be, want_be), what
        assert res["sym_errors"] == int(want_se.sum()) and res["bit_errors"] == int(want_be.sum()), what
    else:
        # a handful of boundary symbols may fall the other way in f32; the tolerance of the north star on rates.
        # A realization with a stream in outage (an eighth or more of its symbols wrong -- one dead stream of up to eight:
        # received at next to no power, every decision a near-tie) gets 2 % of its own error count on top -- found by seed
        # hunts (MCLE_FUZZ_OFFSET=2: 331 vs 327 errors of 387 in one block-diagonalisation realization; OFFSET=1000,
        # 150 trials: 233 vs 222 of 512 in one realization of 1 200 cases, everything else equal).
        outage = want_se >= nsym // 8
        # (an iterative solver in complex64 ends somewhere else altogether in such a realization: 108 against 117 symbol and
        # 172 against 216 bit errors in the OFFSET=7000 case above)
        slack_s = (0.10 if iterative else 0.02) * float(want_se[outage].sum())
        slack_b = (0.25 if iterative else 0.02) * float(want_be[outage].sum())
        assert abs(int(se.sum()) - int(want_se.sum())) <= max(3, 1e-4 * n * nsym) + slack_s, what
        assert abs(int(be.sum()) - int(want_be.sum())) <= max(6, 1e-4 * n * nbits) + slack_b, what


def _snr_for(rs, M):
    return float(rs.uniform(2.0, 10.0) + 3.0 * np.log2(M))


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("trial", range(N_TRIALS))
def test_fuzz_single_carrier(engine, dt, trial):
    rs = np.random.RandomState(100 + trial + 1000 * OFFSET)
    mod, M = MODS[rs.randint(len(MODS))]
    _bind(engine, mod, M)
    N = int(rs.choice([1, 15, 16, 17, 1000, 4099, 16384, 16385, 40000]))
    snr = _snr_for(rs, M)
    first, count = int(rs.randint(0, 1 << 40)), int(rs.randint(1, 6))
    nv = 1.0 / omodem.dB2Linear(snr)
    want = _oracle(chains.chain_awgn, first, count, mod=mod, M=M, N=N, snr_db=snr)
    _check(*engine.run_awgn(N, nv, SEED, first, count, dtype=dt, per_realization=True), want, dt, ("awgn", mod, M, N))
    Fd, Ts, L = float(rs.uniform(5, 300)), float(10 ** rs.uniform(-5, -3)), int(rs.choice([4, 8, 16, 23]))
    want = _oracle(chains.chain_flat_jakes, first, count, mod=mod, M=M, N=N, snr_db=snr, Fd=Fd, Ts=Ts, L=L)
    _check(*engine.run_flat_fading(N, nv, SEED, first, count, Fd=Fd, Ts=Ts, L=L, dtype=dt, per_realization=True), want,
           dt, ("jakes", mod, M, N, Fd, Ts, L))


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("trial", range(N_TRIALS))
def test_fuzz_ofdm_chains(engine, dt, trial):
    rs = np.random.RandomState(200 + trial + 1000 * OFFSET)
    mod, M = MODS[1 + rs.randint(len(MODS) - 1)]
    _bind(engine, mod, M)
    fft = int(rs.choice([64, 128, 256, 512, 1024]))
    cp = int(rs.randint(0, fft // 4))
    used = int(rs.choice([fft, 2 * rs.randint(1, fft // 2)]))
    n_sym = int(rs.randint(1, 4))
    snr = _snr_for(rs, M) + 6.0
    nv = 1.0 / omodem.dB2Linear(snr)
    first, count = int(rs.randint(0, 1 << 33)), int(rs.randint(1, 7))
    # SISO OFDM over a Jakes TDL channel with the one-tap equaliser
    S = int(rs.randint(1, 6))
    delays = tuple(sorted(rs.choice(np.arange(0, min(fft // 4, 24)), size=S, replace=False).tolist()))
    powers = tuple(float(v) for v in -np.sort(rs.uniform(0, 15, size=S)))
    Ts, Fd, L = 1e-6, float(rs.uniform(5, 120)), int(rs.choice([8, 12]))
    kw = dict(mod=mod, M=M, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr, Fd=Fd, Ts=Ts, L=L,
              tap_powers_dB=powers, tap_delays_samples=delays)
    p_lin, d_idx = och.discretize_profile(np.array(powers), np.array(delays) * Ts, Ts)
    want = _oracle(chains.chain_ofdm_tdl, first, count, **kw)
    _check(*engine.run_ofdm_tdl(fft, cp, used, n_sym, nv, p_lin, d_idx, SEED, first, count, Fd=Fd, Ts=Ts, L=L, dtype=dt,
                                per_realization=True), want, dt, ("ofdm_tdl", kw))
    # spatial multiplexing over a flat channel with per-antenna OFDM
    na = int(rs.choice([2, 4]))
    mmse = bool(rs.randint(2))
    kw = dict(mod=mod, M=M, nt=na, nr=na, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr,
              mmse=mmse)
    want = _oracle(chains.chain_mimo_ofdm, first, count, **kw)
    _check(*engine.run_mimo_ofdm(na, na, fft, cp, used, n_sym, nv, SEED, first, count, mmse=mmse, dtype=dt,
                                 per_realization=True), want, dt, ("mimo_ofdm", kw))
    # the same over the frequency-selective channel, one MMSE / ZF filter per subcarrier (fused kernel)
    count = min(count, 3)
    kw = dict(mod=mod, M=M, nt=na, nr=na, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr + 6.0,
              Fd=Fd, Ts=Ts, L=L, tap_powers_dB=powers, tap_delays_samples=delays)
    try:
        got = engine.run_mimo_ofdm_tdl(na, na, fft, cp, used, n_sym, 1.0 / omodem.dB2Linear(snr + 6.0), p_lin, d_idx, SEED,
                                       first, count, Fd=Fd, Ts=Ts, L=L, mmse=True, dtype=dt, per_realization=True)
    except _lib.McleUnsupported:
        return          # Doppler beyond the fused kernel's tap model (the simulator runs the staged chain then)
    _check(*got, _oracle(chains.chain_mimo_ofdm_tdl, first, count, **kw), dt, ("mimo_ofdm_tdl", kw))


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("trial", range(N_MFMA_TRIALS_EARLY))
def test_fuzz_planar_family(engine, dt, trial):
    """The planar MIMO-OFDM kernel family (csrc/pipeline_mimo_planar.hip, both arithmetics): random geometry -- fft_size
    256 .. 2048, any 1 <= Nt <= Nr <= 4 -- cyclic prefix (odd ones take the unpaired noise draws), band, symbol count,
    modulation, demodulator, filter and realization offset, against the oracle on the same draws; where the generic radix-4
    kernel has the shape, against that one too."""
    rs = np.random.RandomState(900 + trial + 1000 * OFFSET)
    mod, M = MODS[1 + rs.randint(len(MODS) - 1)]
    _bind(engine, mod, M)
    fft = int(rs.choice([256, 512, 1024, 1024, 2048]))
    nr = int(rs.randint(1, 4)) + 1
    nt = int(rs.randint(1, nr + 1))
    cp = int(rs.choice([0, 1, 7, 16, 33, rs.randint(0, fft // 4)]))
    used = int(rs.choice([fft, fft, 2 * rs.randint(1, fft // 2), 16 * rs.randint(1, fft // 16)]))
    n_sym = int(rs.randint(1, 4))
    snr = _snr_for(rs, M) + 6.0
    nv = 1.0 / omodem.dB2Linear(snr)
    mmse = bool(rs.randint(2))
    method = _lib.DEMOD_QAM_SLICER if (mod == "qam" and rs.randint(2)) else _lib.DEMOD_MINDIST
    first, count = int(rs.randint(0, 1 << 33)), int(rs.randint(1, 5))
    kw = dict(mod=mod, M=M, nt=nt, nr=nr, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr, mmse=mmse)
    want = _oracle(chains.chain_mimo_ofdm, first, count, **kw)
    run = lambda: engine.run_mimo_ofdm(nt, nr, fft, cp, used, n_sym, nv, SEED, first, count, mmse=mmse, method=method, dtype=dt,
                                       per_realization=True)
    got = run()
    _check(*got, want, dt, ("planar", dt, kw, method))
    if nt == nr and nr in (2, 4) and not (dt == "f64" and (fft, nr) == (2048, 4)):
        with engine.options(f64_generic=1):
            ref = run()
        assert np.max(np.abs(got[1].astype(np.int64) - ref[1].astype(np.int64))) <= (1 if dt == "f64" else 4), ("vs generic", kw)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("trial", range(N_TRIALS))
def test_fuzz_chunked_pipelines(engine, dt, trial):
    rs = np.random.RandomState(300 + trial + 1000 * OFFSET)
    mod, M = MODS[1 + rs.randint(len(MODS) - 1)]
    _bind(engine, mod, M)
    NS = int(rs.choice([2, 7, 64, 100, 128, 129, 200, 500]))
    snr = _snr_for(rs, M) + 4.0
    nv = 1.0 / omodem.dB2Linear(snr)
    first, count = int(rs.randint(0, 1 << 36)), int(rs.randint(1, 70))
    want = _oracle(chains.chain_ia, first, count, mod=mod, M=M, K=3, nr=2, nt=2, Ns=1, NSymbs=NS, snr_db=snr)
    res, se, be, _, _ = engine.run_ia(NS, nv, SEED, first, count, dtype=dt, per_realization=True)
    _check(res, se, be, want, dt, ("ia", mod, M, NS))
    algo = ["alt_min", "min_leakage", "max_sinr", "mmse"][rs.randint(4)]
    init = ["random", "closed_form", "alt_min"][rs.randint(3)] if algo != "alt_min" else "random"
    n_it = min(count, 12)
    kw = dict(algo=algo, mod=mod, M=M, K=3, nr=2, nt=2, Ns=1, NSymbs=NS, snr_db=snr, max_iterations=int(rs.randint(3, 40)),
              initialize_with=init)
    out = [chains.chain_ia_iterative(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + n_it)]
    res, se, be, _, its = engine.run_ia(NS, nv, SEED, first, n_it, dtype=dt, per_realization=True, solver=algo,
                                        max_iterations=kw["max_iterations"], initialize_with=init)
    _check(res, se, be, (np.array([o["symbol_errors"] for o in out]), np.array([o["bit_errors"] for o in out]),
                         out[0]["num_symbols"], out[0]["num_bits"]), dt, ("ia_iterative", kw), iterative=True)
    if dt == "f64":
        assert np.array_equal(its, [o["runned_iterations"] for o in out]), kw
    scheme, nt, nr = [("blast", 2, 3), ("blast", 4, 4), ("mrc", 1, 4), ("mrt", 3, 1), ("alamouti", 2, 2)][rs.randint(5)]
    ns_flat = NS + (NS & 1) if scheme == "alamouti" else NS
    want = _oracle(chains.chain_mimo_scheme, first, count, scheme=scheme, mod=mod, M=M, nt=nt, nr=nr, NSymbs=ns_flat,
                   snr_db=snr)
    _check(*engine.run_mimo_flat(scheme, nt, nr, ns_flat, nv, SEED, first, count, dtype=dt, per_realization=True), want,
           dt, ("flat", scheme, nt, nr, ns_flat))
    K, r = [(2, 1), (3, 1), (2, 2), (3, 2), (4, 2), (2, 3), (2, 4)][rs.randint(7)]
    # A stream the water-filling switches off is received as exactly 0 (the reference: ~1e-17 of rounding residue):
    # its "decision" is a tie between all points of equal modulus, broken by the last bit of |c_m|, i.e. undefined
    # for PSK rings (QAM's innermost points tie exactly and the first wins on both sides).  Keep the water level
    # low for PSK so that every stream stays on.
    bd_nv = float(10 ** (rs.uniform(-6, -3) if mod == "psk" else rs.uniform(-6, 0.3)))
    kw = dict(mod=mod, M=M, K=K, nr=r, NSymbs=NS, iPu=float(rs.uniform(0.5, 3.0)), noise_var=nv,
              bd_noise_var=bd_nv, pathloss=None, waterfill=bool(rs.randint(2)))
    want = _oracle(chains.chain_bd, first, count, canonical=True, **kw)
    _check(*engine.run_bd(K, r, NS, kw["iPu"], nv, SEED, first, count, bd_noise_var=kw["bd_noise_var"],
                          waterfilling=kw["waterfill"], dtype=dt, per_realization=True), want, dt, ("bd", kw))


N_MFMA_TRIALS = int(os.environ.get("MCLE_FUZZ_TRIALS", "10"))


@pytest.mark.parametrize("trial", range(N_MFMA_TRIALS))
def test_fuzz_matrix_core_kernels(engine, trial):
    """f32: the matrix-core kernels of configs 2, 3 and 4 (and, through the last two, fft16.hpp) on random CP lengths,
    band widths, symbol counts, tap sets and realization offsets -- against the oracle on the same draws, and against the
    VALU kernels they replace (engine option no_mfma)."""
    rs = np.random.RandomState(700 + trial + 1000 * OFFSET)
    mod, M = MODS[1 + rs.randint(len(MODS) - 1)]
    _bind(engine, mod, M)
    fft = 1024
    cp = int(rs.choice([0, 1, 7, 16, 33, 72, 255]))
    used = int(rs.choice([fft, fft, 2 * rs.randint(1, fft // 2), 16 * rs.randint(1, 64)]))
    n_sym = int(rs.randint(1, 4))
    snr = _snr_for(rs, M) + 6.0
    nv = 1.0 / omodem.dB2Linear(snr)
    first, count = int(rs.randint(0, 1 << 33)), int(rs.randint(1, 9))

    def both(fn, **matrix_core):
        """the default kernel (configs 2, 3: matrix cores; config 4 since round 4: the planar VALU family) against the round-1
        VALU kernel, and -- config 4 -- the matrix-core kernel (option f32_mfma) against both"""
        got = fn()
        with engine.options(no_mfma=1):
            ref = fn()
        assert np.max(np.abs(got[1].astype(np.int64) - ref[1].astype(np.int64))) <= 4, ("vs the VALU kernel", trial)
        if matrix_core:
            with engine.options(**matrix_core):
                alt = fn()
            assert np.max(np.abs(got[1].astype(np.int64) - alt[1].astype(np.int64))) <= 4, ("planar vs matrix cores", trial)
            assert np.array_equal(got[1] == 0xFFFFFFFF, alt[1] == 0xFFFFFFFF)
        return got

    S = int(rs.randint(1, 6))
    dmax = max(1, min(cp, 24)) if rs.randint(4) else 30          # now and then a delay beyond the CP: the VALU kernel's case
    delays = tuple(sorted(rs.choice(np.arange(0, dmax + 1), size=min(S, dmax + 1), replace=False).tolist()))
    powers = tuple(float(v) for v in -np.sort(rs.uniform(0, 15, size=len(delays))))
    Ts, Fd, L = 1e-6, float(rs.uniform(5, 400)), int(rs.choice([4, 8, 12]))
    kw = dict(mod=mod, M=M, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr, Fd=Fd, Ts=Ts, L=L,
              tap_powers_dB=powers, tap_delays_samples=delays)
    p_lin, d_idx = och.discretize_profile(np.array(powers), np.array(delays) * Ts, Ts)
    want = _oracle(chains.chain_ofdm_tdl, first, count, **kw)
    _check(*both(lambda: engine.run_ofdm_tdl(fft, cp, used, n_sym, nv, p_lin, d_idx, SEED, first, count, Fd=Fd, Ts=Ts, L=L,
                                             dtype="f32", per_realization=True), tdl_kernel=1), want, "f32", ("ofdm_tdl", kw))
    mmse = bool(rs.randint(2))
    method = _lib.DEMOD_QAM_SLICER if (mod == "qam" and rs.randint(2)) else _lib.DEMOD_MINDIST
    kw = dict(mod=mod, M=M, nt=4, nr=4, fft_size=fft, cp_size=cp, num_used=used, n_ofdm_sym=n_sym, snr_db=snr, mmse=mmse)
    want = _oracle(chains.chain_mimo_ofdm, first, count, **kw)
    _check(*both(lambda: engine.run_mimo_ofdm(4, 4, fft, cp, used, n_sym, nv, SEED, first, count, mmse=mmse, method=method,
                                              dtype="f32", per_realization=True), f32_mfma=1), want, "f32", ("mimo_ofdm", kw))
    # config 2 (k_run_flat_mfma: 8 / 16 rays): any length (ragged quads, groups, chunks), sampling time and Doppler
    N = int(rs.choice([rs.randint(1, 70), rs.randint(70, 5000), 16384 + rs.randint(0, 3000)]))
    Lf, Tsf, Fdf = int(rs.choice([8, 16])), float(10.0 ** rs.uniform(-5, -2.5)), float(rs.uniform(1, 300))
    method = _lib.DEMOD_QAM_SLICER if (mod == "qam" and rs.randint(2)) else _lib.DEMOD_MINDIST
    kw = dict(mod=mod, M=M, N=N, snr_db=snr - 6.0, Fd=Fdf, Ts=Tsf, L=Lf)
    want = _oracle(chains.chain_flat_jakes, first, count, **kw)
    _check(*both(lambda: engine.run_flat_fading(N, 1.0 / omodem.dB2Linear(snr - 6.0), SEED, first, count, Fd=Fdf, Ts=Tsf,
                                                L=Lf, method=method, dtype="f32", per_realization=True)),
           want, "f32", ("flat_jakes", kw))


@pytest.mark.parametrize("trial", range(N_MFMA_TRIALS))
def test_fuzz_ofdm_matrix_core_operators(engine, trial):
    """complex64 OFDM modulate / demodulate at 1024 points (kernels_ofdm_mfma.hip) on random batches, symbol counts (any
    remainder modulo the four symbols of a pass), zero-padded tails, CP lengths and band widths."""
    from oracle import ofdm as oofdm
    rs = np.random.RandomState(900 + trial + 1000 * OFFSET)
    batch, n_sym = int(rs.randint(1, 6)), int(rs.randint(1, 12))
    cp = int(rs.choice([0, 1, 16, 37, 72, 511, 1024]))
    used = int(rs.choice([1024, 2 * rs.randint(1, 512)]))
    n_in = n_sym * used - int(rs.randint(0, used))
    x = (rs.randn(batch, n_in) + 1j * rs.randn(batch, n_in)).astype(np.complex64)
    tx = engine.ofdm_modulate(x, 1024, cp, used, batch=batch, dtype="f32")
    want = np.stack([oofdm.modulate(x[b].astype(complex), 1024, cp, used) for b in range(batch)])
    err = np.max(np.abs(tx - want)) / max(1e-30, np.max(np.abs(want)))
    assert tx.shape == want.shape and err <= 3e-6, (batch, n_sym, cp, used, n_in, err)
    back = engine.ofdm_demodulate(want, 1024, cp, used, batch=batch, dtype="f32")
    wantb = np.stack([oofdm.demodulate(want[b], 1024, cp, used) for b in range(batch)])
    errb = np.max(np.abs(back - wantb)) / max(1e-30, np.max(np.abs(wantb)))
    assert back.shape == wantb.shape and errb <= 3e-6, (batch, n_sym, cp, used, n_in, errb)
