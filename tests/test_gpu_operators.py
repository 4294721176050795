"""GPU: every per-operator HIP kernel (through the C ABI) against the reference-minted golden
vectors and the NumPy oracle.  f64 instantiation: integers bit-exact, floats <= 1e-11 relative;
f32 instantiation: floats <= 2e-5 relative, decisions allowed to differ only on boundary cases."""
import math

import numpy as np
import pytest

from helpers import golden_cases, relerr
from oracle import chains, channels as och, modem as omodem, ofdm as oofdm, philox as P
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu

F64_TOL, F32_TOL = 1e-11, 2e-5
TOL = {"f64": F64_TOL, "f32": F32_TOL}


@pytest.mark.parametrize("M", [4, 16, 64, 256])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_modulate_demodulate_qam(engine, golden_ops, M, dt):
    table = golden_ops["qam%d" % M]
    rx, idx, dec = (golden_ops["demod_qam%d_%s" % (M, k)] for k in ("rx", "idx", "dec"))
    engine.set_constellation(table, _lib.CONST_QAM)
    tx = engine.modulate(idx, dtype=dt)
    assert relerr(tx, table[idx]) <= (0 if dt == "f64" else 1e-7)
    got = engine.demodulate(rx, dtype=dt)
    sl = engine.demodulate(rx, method=_lib.DEMOD_QAM_SLICER, dtype=dt)
    if dt == "f64":
        assert np.array_equal(got, dec) and np.array_equal(sl, dec)
    else:
        assert np.count_nonzero(got != dec) <= 2 and np.count_nonzero(sl != dec) <= 2
    assert got.dtype == np.int64 and got.shape == dec.shape
    cnt, se, be = engine.count_errors(idx, dec, int(np.log2(M)))
    assert int(se[0]) == int(np.sum(idx != dec)) and int(be[0]) == int(golden_ops["demod_qam%d_biterr" % M])
    assert cnt["sym_errors"] == int(se[0]) and cnt["bit_errors_sq"] == int(be[0]) ** 2
    cnt2, se2, be2 = engine.demod_count(rx, idx, dtype="f64")
    assert (int(se2[0]), int(be2[0])) == (int(se[0]), int(be[0]))


def test_demodulate_psk_and_bpsk(engine, golden_ops):
    engine.set_constellation(golden_ops["psk8"])
    assert np.array_equal(engine.demodulate(golden_ops["demod_psk8_rx"]), golden_ops["demod_psk8_dec"])
    engine.set_constellation(golden_ops["bpsk"], _lib.CONST_BPSK)
    rx = np.array([0.3, -0.2, 1.7, -4.0, 0.0]) + 0j
    assert list(engine.demodulate(rx)) == [0, 1, 0, 1, 0]
    assert np.array_equal(engine.modulate(np.array([0, 1, 1, 0])), np.array([1, -1, -1, 1]) + 0j)


def test_modulate_errors_and_edges(engine, golden_ops):
    engine.set_constellation(golden_ops["qam16"], _lib.CONST_QAM)
    with pytest.raises(ValueError):            # fundamental.py:196-199
        engine.modulate(np.array([0, 16, 3]))
    assert np.array_equal(engine.modulate(np.array([-1])), golden_ops["qam16"][[-1]])   # negatives wrap
    assert engine.modulate(np.zeros(0, dtype=int)).shape == (0,)
    assert engine.demodulate(np.zeros((0,), dtype=complex)).shape == (0,)
    assert engine.demodulate(golden_ops["qam16"].reshape(4, 4)).shape == (4, 4)
    engine.set_constellation(golden_ops["psk8"])
    with pytest.raises(_lib.McleError):        # slicer needs a QAM table
        engine.demodulate(np.ones(4, dtype=complex), method=_lib.DEMOD_QAM_SLICER)
    with pytest.raises(_lib.McleError):        # and a QAM claim is verified against the reference layout
        engine.set_constellation(golden_ops["psk16"], _lib.CONST_QAM)


def test_count_errors_ragged_batches(engine):
    rs = np.random.RandomState(3)
    for n_real, n in ((1, 1), (7, 13), (300, 257), (3, 100003)):
        a = rs.randint(0, 64, (n_real, n))
        b = np.where(rs.rand(n_real, n) < 0.1, rs.randint(0, 64, (n_real, n)), a)
        cnt, se, be = engine.count_errors(a, b, 6, n_real)
        want_s = (a != b).sum(axis=1)
        want_b = omodem.count_bit_errors(a, b, 1)
        assert np.array_equal(se, want_s) and np.array_equal(be, want_b)
        assert cnt["sym_errors"] == want_s.sum() and cnt["sym_errors_sq"] == int((want_s.astype(np.int64) ** 2).sum())
        assert cnt["bit_errors"] == want_b.sum() and cnt["n_realizations"] == n_real
        assert cnt["n_symbols"] == n and cnt["n_bits"] == 6 * n


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_philox_draws(engine, dt):
    seed, r = 20260927, 123456789012
    z = engine.randn_c(1001, seed, r, _lib.STREAM_NOISE, first=5, variance=0.25, dtype=dt)
    want = 0.5 * P.cnormal(seed, r, 1001, P.STREAM_NOISE, offset=5)
    assert relerr(z, want) <= (1e-13 if dt == "f64" else 3e-6)
    s = engine.rand_symbols(999, 64, seed, r, first=3)
    assert np.array_equal(s, P.symbols(seed, r, 999, 64, offset=3))
    # the generators take one Philox block per thread: every alignment of the window against the blocks
    for first in (0, 1, 2, 7, 16, 21):
        for n in (1, 2, 3, 15, 16, 17, 33, 4097):
            z = engine.randn_c(n, seed, r, _lib.STREAM_CHAN, first=first, dtype=dt)
            assert relerr(z, P.cnormal(seed, r, n, P.STREAM_CHAN, offset=first)) <= (1e-13 if dt == "f64" else 3e-6), (first, n)
            if dt == "f64":
                assert np.array_equal(engine.rand_symbols(n, 16, seed, r, first=first), P.symbols(seed, r, n, 16, offset=first))
    if dt == "f64":
        rows = engine.rand_symbols_batch(37, 8, seed, r, 3).get()
        for k in range(3):
            assert np.array_equal(rows[k], P.symbols(seed, r + k, 37, 8))


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("mod,M", [("qam", 64), ("psk", 8), ("bpsk", 2)])
def test_rand_modulate_batch_equals_draw_then_modulate(engine, dt, mod, M):
    """The one-pass "gen + modulate" operator of the staged chains: labels = the oracle's DATA draws, samples = the
    table looked up at them (modulators/fundamental.py:175-199), for rows with and without whole 16-byte groups."""
    from oracle import chains
    table = chains.constellation(mod, M)
    engine.set_constellation(table, _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC)
    seed, first = 99, (1 << 33) + 5
    for n in (4096, 37, 16, 5):
        idx, sym = engine.rand_modulate_batch(n, seed, first, 3, dtype=dt)
        idx, sym = idx.get(), sym.get()
        for k in range(3):
            want = P.symbols(seed, first + k, n, M)
            assert np.array_equal(idx[k], want)
            assert np.array_equal(sym[k], table[want].astype(sym.dtype))


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_byte_label_operators_equal_the_int32_ones(engine, dt):
    """mcle_rand_modulate_batch_u8 / mcle_demod_count_u8 (SURVEY 8(d)'s I = 1 B): the same labels, samples and error counts
    as the int32 forms, for rows with and without whole 16-label groups."""
    import numpy as np
    from oracle import chains
    table = chains.constellation("qam", 64)
    engine.set_constellation(table, _lib.CONST_QAM)
    seed, first = 5, (1 << 34) + 3
    rs = np.random.RandomState(1)
    for n in (4096, 48, 37, 5):
        i32, s32 = engine.rand_modulate_batch(n, seed, first, 3, dtype=dt)
        i8, s8 = engine.rand_modulate_batch(n, seed, first, 3, dtype=dt, labels=np.uint8)
        assert i8.dtype == np.uint8 and np.array_equal(i8.get().astype(np.int32), i32.get())
        assert np.array_equal(s8.get(), s32.get())
        noisy = s32.get() + (0.12 * (rs.randn(3, n) + 1j * rs.randn(3, n))).astype(s32.dtype)
        rx = engine.to_device(noisy, s32.dtype)
        c32 = engine.demod_count(rx, i32, n_real=3, dtype=dt)
        c8 = engine.demod_count(rx, i8, n_real=3, dtype=dt)
        assert c32[0] == c8[0] and np.array_equal(c32[1], c8[1]) and np.array_equal(c32[2], c8[2])
        assert c32[0]["sym_errors"] > 0 or n < 40


def test_rand_modulate_batch_refuses_tables_beyond_a_byte(engine):
    """Labels are Philox bytes: a 512-point table is outside the operator's envelope (as in mcle_rand_symbols_batch) and must
    be an error, not wrong labels (the byte mask overflowed silently for M = 512 / 1024)."""
    import numpy as np
    table = np.exp(2j * np.pi * np.arange(512) / 512)
    engine.set_constellation(table, _lib.CONST_GENERIC)
    with pytest.raises((_lib.McleError, ValueError), match="power of two in"):
        engine.rand_modulate_batch(64, 1, 0, 2, dtype="f64")
    with pytest.raises((_lib.McleError, ValueError)):
        engine.rand_symbols_batch(64, 512, 1, 0, 2)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_awgn_chain_injected(engine, dt):
    """C1 with the reference's own draws injected: decisions / counters bit-exact (f64)."""
    for kw, reals in golden_cases("c1_awgn"):
        for g in reals:
            if "noise" not in g:
                continue
            kind = _lib.CONST_QAM if kw["mod"] == "qam" else _lib.CONST_GENERIC
            engine.set_constellation(g["table"], kind)
            tx = engine.modulate(g["idx"], dtype=dt)
            rx = engine.awgn_add(tx, g["noise"], float(g["noise_var"]), dtype=dt)
            assert relerr(rx, g["rx"]) <= TOL[dt]
            dec = engine.demodulate(rx, dtype=dt)
            cnt, se, be = engine.count_errors(g["idx"], dec, omodem.level2bits(kw["M"]))
            if dt == "f64":
                assert np.array_equal(dec, g["decisions"])
                assert (int(se[0]), int(be[0])) == (int(g["symbol_errors"]), int(g["bit_errors"]))
            else:
                assert abs(int(se[0]) - int(g["symbol_errors"])) <= 2


@pytest.mark.parametrize("fft,cp,used", [(16, 4, 10), (64, 16, 52), (64, 0, 64), (1024, 16, 1024), (1024, 72, 600)])
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_ofdm_mod_demod(engine, golden_ops, fft, cp, used, dt):
    key = "ofdm_%d_%d_%d" % (fft, cp, used)
    x, tx, back = golden_ops[key + "_x"], golden_ops[key + "_tx"], golden_ops[key + "_back"]
    got = engine.ofdm_modulate(x, fft, cp, used, dtype=dt).reshape(-1)
    assert got.shape == tx.shape and relerr(got, tx) <= TOL[dt]
    got_b = engine.ofdm_demodulate(tx, fft, cp, used, dtype=dt).reshape(-1)
    assert got_b.shape == back.shape and relerr(got_b, back) <= TOL[dt] * 10
    # round trip recovers the (zero padded) input: reference tests/modulators_package_test.py:606-650
    assert relerr(got_b[:x.size], x) <= TOL[dt] * 10 and np.max(np.abs(got_b[x.size:])) <= TOL[dt] * 10


def test_ofdm_sizes_batches_and_errors(engine):
    rs = np.random.RandomState(5)
    for fft in (16, 32, 128, 256, 512, 2048, 4096):
        x = rs.randn(3, fft * 2) + 1j * rs.randn(3, fft * 2)
        tx = engine.ofdm_modulate(x, fft, fft // 8, fft, batch=3)
        want = np.stack([oofdm.modulate(x[b], fft, fft // 8, fft) for b in range(3)])
        assert relerr(tx, want) <= F64_TOL
        back = engine.ofdm_demodulate(tx, fft, fft // 8, fft, batch=3)
        assert relerr(back, x) <= F64_TOL * 10
    with pytest.raises(ValueError):
        engine.ofdm_modulate(np.ones(8, dtype=complex), 16, 17, 16)      # ofdm.py:75-78
    with pytest.raises(ValueError):
        engine.ofdm_modulate(np.ones(8, dtype=complex), 16, 4, 18)       # ofdm.py:83-86
    with pytest.raises(ValueError):
        engine.ofdm_modulate(np.ones(8, dtype=complex), 16, 4, 7)        # ofdm.py:87-90


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_ofdm_any_fft_size(engine, dt):
    """np.fft takes any length (the reference's equaliser test uses 24 subcarriers,
    tests/modulators_package_test.py:659-712): non power-of-two sizes -- composite, prime, tiny, the
    LTE 1536 grid, the largest odd size -- run the two-pass DFT kernels and agree with the oracle."""
    rs = np.random.RandomState(11)
    tol = F64_TOL * 50 if dt == "f64" else 2e-5
    for fft, cp, used in ((24, 8, 24), (2, 0, 2), (8, 2, 6), (12, 3, 10), (97, 10, 60), (600, 40, 512),
                          (1536, 108, 1200), (4095, 17, 4000), (3000, 0, 3000)):
        n_in = 2 * used + used // 2            # third symbol is zero padded
        x = rs.randn(2, n_in) + 1j * rs.randn(2, n_in)
        tx = engine.ofdm_modulate(x, fft, cp, used, batch=2, dtype=dt)
        want = np.stack([oofdm.modulate(x[b], fft, cp, used) for b in range(2)])
        assert tx.shape == want.shape and relerr(tx, want) <= tol, (fft, relerr(tx, want))
        back = engine.ofdm_demodulate(want, fft, cp, used, batch=2, dtype=dt)
        wantb = np.stack([oofdm.demodulate(want[b], fft, cp, used) for b in range(2)])
        assert relerr(back, wantb) <= tol, (fft, relerr(back, wantb))
        assert relerr(back[:, :n_in], x) <= tol * 10
    # one-tap equaliser and mean responses on the 24-subcarrier grid of the reference's test
    fft, cp, used, n_sym = 24, 8, 24, 3
    delays = np.array([0, 1, 3, 5], dtype=np.int32)
    taps = (rs.randn(4, n_sym * (fft + cp)) + 1j * rs.randn(4, n_sym * (fft + cp))) * 0.5
    data = rs.randn(n_sym * used) + 1j * rs.randn(n_sym * used)
    got = engine.onetap_equalize(data, taps, delays, fft, cp, used, dtype=dt)
    want = oofdm.onetap_equalize(data, taps, delays, fft, cp, used)
    assert relerr(got, want) <= tol * 20
    H = engine.tdl_mean_freq_response(taps, delays, n_sym, fft, cp, used, dtype=dt)
    idx = oofdm.used_subcarrier_indexes(fft, used)
    for s in range(n_sym):
        mean = taps[:, s * (fft + cp):(s + 1) * (fft + cp)].mean(axis=1)
        full = np.zeros(fft, dtype=complex)
        full[delays] = mean
        assert relerr(H[s], np.fft.fft(full)[idx]) <= tol * 20


def test_equalizer_app_of_the_reference_on_24_subcarriers(engine):
    """The reference's OfdmOneTapEqualizer test as an application of the mirror classes
    (tests/modulators_package_test.py:659-712): noiseless QPSK over a Jakes/TU channel, 24-point OFDM,
    one-tap equalisation recovers every symbol."""
    from pyphysim_amd import channels, modulators
    rs = np.random.RandomState(3)
    n_sc = 24
    ofdm_obj = modulators.OFDM(n_sc, cp_size=8, engine=engine)
    eq = modulators.OfdmOneTapEqualizer(ofdm_obj)
    qam = modulators.QAM(4, engine=engine)
    data = rs.randint(0, 4, size=2 * n_sc)
    tx = ofdm_obj.modulate(qam.modulate(data))
    jakes = channels.JakesSampleGenerator(50, 1.0 / (55e3 * n_sc), 16, shape=None, RS=rs, engine=engine)
    tdl = channels.TdlChannel(jakes, tap_powers_dB=channels.COST259_TUx.tap_powers_dB,
                              tap_delays=channels.COST259_TUx.tap_delays)
    memory = tdl.num_taps_with_padding - 1
    rx = tdl.corrupt_data(tx)
    got = eq.equalize_data(ofdm_obj.demodulate(rx[:-memory]), tdl.get_last_impulse_response())
    assert np.array_equal(qam.demodulate(got), data)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_rayleigh_injected(engine, dt):
    """a7 staged: the reference's own h and noise (tests/golden/c2b_flat_rayleigh.npz, both of its statements of
    the chain) through modulate -> TDL apply (one 0-delay tap) -> AWGN -> divide -> demodulate."""
    for kw, reals in golden_cases("c2b_flat_rayleigh"):
        kind = {"qam": _lib.CONST_QAM, "bpsk": _lib.CONST_BPSK}.get(kw["mod"], _lib.CONST_GENERIC)
        for g in reals:
            engine.set_constellation(g["table"], kind)
            h = g["h"].reshape(-1)
            tx = engine.modulate(g["idx"], dtype=dt)
            faded = engine.tdl_apply(tx, h, [0], dtype=dt)
            assert relerr(faded, h * g["table"][g["idx"]]) <= (1e-12 if dt == "f64" else 1e-6)
            rx = engine.awgn_add(faded, g["noise"], float(g["noise_var"]), dtype=dt)
            eq = engine.cdiv(rx, h, dtype=dt)
            # a deep fade amplifies rounding: compare where |h| is not tiny
            ok = np.abs(h) > 1e-2
            assert relerr(eq[ok], g["eq"][ok]) <= (1e-9 if dt == "f64" else 2e-3)
            dec = engine.demodulate(eq, dtype=dt)
            if dt == "f64":
                assert np.array_equal(dec, g["decisions"])
                assert int(np.sum(dec != g["idx"])) == int(g["symbol_errors"])
            else:
                assert np.count_nonzero(dec != g["decisions"]) <= 3


def test_rayleigh_class_mirror_draws(engine):
    """channels.RayleighSampleGenerator / SuChannel (reference fading_generators.py:208-282, singleuser.py:48-81):
    the ctor consumes one sample, corrupt_data draws N more CN(0,1) values -- here positions 1..N of the NOISE
    stream selected by util.seed -- and multiplies them onto the signal."""
    from oracle import philox
    from pyphysim_amd import channels, util
    n = 4096
    util.seed(99, 7)
    chan = channels.SuChannel(channels.RayleighSampleGenerator())
    tx = np.exp(1j * np.linspace(0, 20, n))
    out = chan.corrupt_data(tx)
    h = chan.get_last_impulse_response().tap_values_sparse
    want = philox.cnormal(99, 7, n, philox.STREAM_NOISE, offset=1)
    assert h.shape == (1, n) and relerr(h[0], want) <= 1e-12
    assert relerr(out, want * tx) <= 1e-12
    assert abs(np.mean(np.abs(h) ** 2) - 1.0) < 0.05


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_blast_filter_staged_equals_the_direct_kernel(engine, dt):
    """k_blast_filter_staged (>= 64 matrices, 16-byte multiples: coalesced through LDS) gives the values of k_blast_filter
    (smaller batches) bit for bit, ragged batch tails included."""
    rs = np.random.RandomState(77)
    cdt = np.complex128 if dt == "f64" else np.complex64
    for nr, nt in ((2, 2), (4, 2), (4, 4), (3, 2), (2, 1), (3, 3)):
        for batch in (64, 131, 257):
            H = (rs.randn(batch, nr, nt) + 1j * rs.randn(batch, nr, nt)).astype(cdt)
            H[5] = 0                                            # a singular channel: flagged, not NaN-poisoning its neighbours
            for nv in (0.0, 0.05):
                G, sk = engine.blast_filter(H, nv, dtype=dt)
                parts = [engine.blast_filter(H[i:i + 50], nv, dtype=dt) for i in range(0, batch, 50)]
                Gp, skp = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
                assert np.array_equal(sk, skp), (nr, nt, batch, nv)
                good = sk == 0
                assert np.array_equal(G[good], Gp[good]), (nr, nt, batch, nv)
                assert (nv > 0) or sk[5] == 1


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_jakes_block_kernel_equals_the_direct_sum(engine, dt):
    """k_jakes_blocks (uniform time axis, L <= 16, >= 1024 samples: one phasor per ray and 64-sample block times the lane's
    rotation) against the closed form sample by sample (NumPy, fading_generators.py:519-522) and against k_jakes
    (engine option jakes_direct), for ray counts around the four instantiations, ragged lengths and late start times."""
    rs = np.random.RandomState(31)
    tol = 1e-9 if dt == "f64" else 3e-5
    for L, S, n, t0, step, Fd in ((8, 1, 100000, 1e-3, 1e-3, 100.0), (1, 3, 1024, 0.0, 1e-4, 30.0), (5, 2, 4097, 7.5, 1e-3, 250.0),
                                  (12, 4, 70001, 1e-3, 2e-5, 900.0), (16, 2, 65536, 123.0, 1e-3, 100.0), (13, 1, 5000, 0.0, 1e-2, 5.0),
                                  (20, 2, 3000, 0.0, 1e-3, 100.0)):          # L = 20: outside the block kernel, k_jakes runs
        phi, psi = rs.uniform(0, 2 * np.pi, (L, S)), rs.uniform(0, 2 * np.pi, (L, S))
        t = t0 + np.arange(n) * step
        want = och.jakes_samples(phi[:, :, None], psi[:, :, None], Fd, t)
        pw = rs.uniform(0.1, 1.0, S)
        h = engine.jakes_generate(phi, psi, Fd, t0, step, n, tap_power=pw, dtype=dt)
        assert h.shape == (S, n)
        assert np.max(np.abs(h - np.sqrt(pw)[:, None] * want)) <= tol * max(1.0, 1e3 * abs(t0)), (L, S, n)
        with engine.options(jakes_direct=1):
            ref = engine.jakes_generate(phi, psi, Fd, t0, step, n, tap_power=pw, dtype=dt)
        assert np.max(np.abs(h - ref)) <= tol * max(1.0, 1e3 * abs(t0)), (L, S, n)
        if L > 16:
            assert np.array_equal(h, ref)


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_jakes_tdl_equalizer_injected(engine, dt):
    """C2 / C3 with the reference's phi, psi and noise injected."""
    for kw, reals in golden_cases("c2_flat_jakes"):
        for g in reals:
            if "noise" not in g:
                continue
            n = g["idx"].size
            t, _ = och.jakes_time_axis(kw["Ts"], kw["Ts"], n)
            dt_step = float(t[1] - t[0])
            phi, psi = g["phi"].reshape(kw["L"], 1), g["psi"].reshape(kw["L"], 1)
            h = engine.jakes_generate(phi, psi, kw["Fd"], kw["Ts"], dt_step, n, dtype=dt)
            assert relerr(h, g["h"]) <= (1e-9 if dt == "f64" else 3e-5)
            engine.set_constellation(g["table"], _lib.CONST_QAM)
            tx = engine.modulate(g["idx"], dtype=dt)
            faded = engine.tdl_apply(tx, h, [0], dtype=dt)
            rx = engine.awgn_add(faded, g["noise"], float(g["noise_var"]), dtype=dt)
            eq = engine.cdiv(rx, h.reshape(-1), dtype=dt)
            assert relerr(eq, g["eq"]) <= (1e-8 if dt == "f64" else 2e-3)
            dec = engine.demodulate(eq, dtype=dt)
            if dt == "f64":
                assert np.array_equal(dec, g["decisions"])
            else:
                assert np.count_nonzero(dec != g["decisions"]) <= 3
    for kw, reals in golden_cases("c3_ofdm_tdl"):
        for g in reals:
            if "taps" not in g or "noise" not in g:
                continue                       # large arrays are stored for one realization only
            fft, cp = kw["fft_size"], kw["cp_size"]
            used = kw["num_used"] or fft
            n = g["tx"].size
            S = len(g["delay_indexes"])
            t, _ = och.jakes_time_axis(kw["Ts"], kw["Ts"], n)
            phi, psi = g["phi"].reshape(kw["L"], S), g["psi"].reshape(kw["L"], S)
            taps = engine.jakes_generate(phi, psi, kw["Fd"], kw["Ts"], float(t[1] - t[0]), n,
                                         tap_power=g["tap_powers_linear"], dtype=dt)
            assert relerr(taps, g["taps"]) <= (1e-9 if dt == "f64" else 3e-5)
            faded = engine.tdl_apply(g["tx"], taps, g["delay_indexes"], dtype=dt)
            assert faded.size == n + int(g["delay_indexes"][-1])
            rx = engine.awgn_add(faded, g["noise"], float(g["noise_var"]), dtype=dt)
            assert relerr(rx, g["rx"]) <= (1e-9 if dt == "f64" else 3e-5)
            demod = engine.ofdm_demodulate(rx[:n], fft, cp, used, dtype=dt).reshape(-1)
            assert relerr(demod, g["demod"]) <= (1e-9 if dt == "f64" else 1e-4)
            eq = engine.onetap_equalize(demod, taps, g["delay_indexes"], fft, cp, used, dtype=dt)
            assert relerr(eq, g["eq"]) <= (1e-8 if dt == "f64" else 2e-3)
            engine.set_constellation(g["table"])
            dec = engine.demodulate(eq, dtype=dt)
            if dt == "f64":
                assert np.array_equal(dec, g["decisions"])
            else:
                assert np.count_nonzero(dec != g["decisions"]) <= 3


def test_tdl_matches_explicit_shifted_sum(engine):
    """Property test of reference tests/channels_package_test.py:821-890 on random taps."""
    rs = np.random.RandomState(11)
    n, delays = 777, np.array([0, 3, 4, 19])
    x = rs.randn(n) + 1j * rs.randn(n)
    g = rs.randn(4, n) + 1j * rs.randn(4, n)
    want = och.tdl_apply(x, g, delays)
    assert relerr(engine.tdl_apply(x, g, delays), want) <= 1e-14


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_blast_injected(engine, golden_ops, dt):
    H, x, nv = golden_ops["blast_H"], golden_ops["blast_x"], float(golden_ops["blast_nv"])
    enc = engine.blast_encode(x, 4, dtype=dt)[0]
    assert relerr(enc, golden_ops["blast_enc"]) <= (0 if dt == "f64" else 1e-7)
    y = engine.mimo_channel(H[None], enc[None], dtype=dt)[0]
    assert relerr(y, golden_ops["blast_y"]) <= TOL[dt]
    for var, key in ((0.0, "blast_zf"), (nv, "blast_mmse")):
        G, skipped = engine.blast_filter(H[None], var, dtype=dt)
        assert skipped[0] == 0
        est = engine.blast_decode(G, y[None], dtype=dt)[0]
        assert relerr(est, golden_ops[key]) <= (1e-10 if dt == "f64" else 2e-4)
    with pytest.raises(ValueError):                       # mimo.py:633-637
        engine.blast_encode(x[:7], 4)
    G, skipped = engine.blast_filter(np.ones((1, 4, 4), dtype=complex), 0.0)   # rank 1 -> flagged
    assert skipped[0] == 1


def test_blast_shapes(engine):
    rs = np.random.RandomState(2)
    from oracle import mimo as omimo
    for nr, nt in ((1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 2), (4, 3), (4, 4)):
        H = (rs.randn(5, nr, nt) + 1j * rs.randn(5, nr, nt)) / np.sqrt(2)
        for nv in (0.0, 0.1):
            G, sk = engine.blast_filter(H, nv)
            want = np.stack([omimo.blast_receive_filter(H[b], nv) for b in range(5)])
            assert relerr(G, want) <= 1e-10 and not sk.any()


@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_mimo_ofdm_chain_injected(engine, dt):
    """C4 staged operator by operator with the reference's H, data and noise injected."""
    for kw, reals in golden_cases("c4_mimo_ofdm"):
        for g in reals:
            if "noise" not in g or "T" not in g:
                continue
            nt, nr, fft, cp = kw["nt"], kw["nr"], kw["fft_size"], kw["cp_size"]
            used = kw["num_used"] or fft
            engine.set_constellation(g["table"], _lib.CONST_QAM)
            sym = engine.modulate(g["idx"], dtype=dt)
            X = engine.blast_encode(sym, nt, dtype=dt)[0]
            T = engine.ofdm_modulate(X, fft, cp, used, batch=nt, dtype=dt)
            assert relerr(T, g["T"]) <= TOL[dt]
            R = engine.mimo_channel(g["H"][None], T[None], g["noise"][None], float(g["noise_var"]), dtype=dt)[0]
            Y = engine.ofdm_demodulate(R, fft, cp, used, batch=nr, dtype=dt)
            assert relerr(Y, g["Y"]) <= TOL[dt] * 10
            G, sk = engine.blast_filter(g["H"][None], float(g["noise_var"]) if kw["mmse"] else 0.0, dtype=dt)
            assert relerr(G[0], g["G"]) <= (1e-9 if dt == "f64" else 5e-4)
            est = engine.blast_decode(G, Y[None], dtype=dt)[0]
            assert relerr(est, g["est"]) <= (1e-9 if dt == "f64" else 2e-3)
            dec = engine.demodulate(est, dtype=dt)
            if dt == "f64":
                assert np.array_equal(dec, g["decisions"])
            else:
                assert np.count_nonzero(dec != g["decisions"]) <= 4


# ---- a13: Alamouti, MRT, SVDMimo (reference mimo/mimo.py:666-1287) --------------------------------
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_alamouti_mrt_injected(engine, golden_ops, dt):
    g = golden_ops
    enc = engine.alamouti_encode(g["ala_x"], dtype=dt)[0]
    assert relerr(enc, g["ala_enc"]) <= (0 if dt == "f64" else 1e-7)
    dec = engine.alamouti_decode(g["ala_H"][None], g["ala_y"][None], dtype=dt)[0]
    assert relerr(dec, g["ala_dec"]) <= (1e-13 if dt == "f64" else 2e-6)
    clean = engine.alamouti_decode(g["ala_H"][None], (g["ala_H"] @ g["ala_enc"])[None], dtype=dt)[0]
    assert relerr(clean, g["ala_x"]) <= (1e-12 if dt == "f64" else 1e-5)      # tests/mimo_package_test.py:610-637
    em = engine.mrt_encode(g["mrt_h"][None], g["mrt_x"][None], dtype=dt)[0]
    assert relerr(em, g["mrt_enc"]) <= (1e-15 if dt == "f64" else 1e-6)
    dm = engine.mrt_decode(g["mrt_h"][None], g["mrt_y"][None], dtype=dt)[0]
    assert relerr(dm, g["mrt_dec"]) <= (1e-14 if dt == "f64" else 1e-6)
    with pytest.raises(ValueError):
        engine.alamouti_encode(np.ones(3, dtype=complex))


def test_svd_filters(engine, golden_ops):
    """Singular values equal LAPACK's; W and G form a consistent pair: W unitary / sqrt(n), G H W = I,
    so SVDMimo.decode(H @ SVDMimo.encode(x)) == x (reference tests/mimo_package_test.py SVD cases)."""
    rs = np.random.RandomState(8)
    for n in (2, 3, 4):
        H = (rs.randn(6, n, n) + 1j * rs.randn(6, n, n)) / np.sqrt(2)
        W, G, S = engine.svd_filters(H)
        for b in range(6):
            assert relerr(S[b], np.linalg.svd(H[b])[1]) <= 1e-12
            assert relerr(W[b].conj().T @ W[b] * n, np.eye(n)) <= 1e-12
            assert relerr(G[b] @ H[b] @ W[b], np.eye(n)) <= 1e-10
            assert relerr(G[b] @ G[b].conj().T, n * np.diag(1.0 / S[b] ** 2)) <= 1e-9
    from pyphysim_amd.mimo import SVDMimo
    m = SVDMimo(golden_ops["svd_H"], engine=engine)
    x = golden_ops["svd_x"]
    assert relerr(m.decode(golden_ops["svd_H"] @ m.encode(x)), x) <= 1e-10
    assert relerr(golden_ops["svd_dec"], x) <= 1e-10                               # the reference recovers x too
    assert relerr(engine.svd_filters(golden_ops["svd_H"][None])[2][0], golden_ops["svd_S"]) <= 1e-12


def test_host_mirror_classes(engine, golden_ops):
    """The drop-in classes (same names / arguments / exceptions as the reference) on the HIP path."""
    from pyphysim_amd import channels, mimo, modulators, util
    q = modulators.QAM(16, engine=engine)
    assert q.name == "16-QAM" and q.M == 16 and q.K == 4
    idx = golden_ops["demod_qam16_idx"]
    assert np.array_equal(q.modulate(idx), golden_ops["qam16"][idx])
    assert np.array_equal(q.demodulate(golden_ops["demod_qam16_rx"]), golden_ops["demod_qam16_dec"])
    assert np.array_equal(q.demodulate(golden_ops["demod_qam16_rx"], method=_lib.DEMOD_QAM_SLICER),
                          golden_ops["demod_qam16_dec"])
    with pytest.raises(ValueError):
        q.modulate(np.array([16]))
    with pytest.raises(ValueError):
        modulators.QAM(32)
    # theory curves: values computed with the reference's own classes (fundamental.py:462-501,780-857)
    assert abs(q.calcTheoreticalSER(10.0) - 0.22203085027243796) < 1e-15
    assert abs(q.calcTheoreticalBER(10.0) - 0.058987202643856936) < 1e-15
    assert abs(modulators.PSK(8, engine=engine).calcTheoreticalSER(10.0) - 0.08700502129401143) < 1e-15
    assert abs(modulators.PSK(8, engine=engine).calcTheoreticalBER(10.0) - 0.029001673764670475) < 1e-15
    assert abs(modulators.BPSK(engine=engine).calcTheoreticalSER(3.0) - 0.022878407561085334) < 1e-15
    q64 = modulators.QAM(64, engine=engine)
    assert abs(q64.calcTheoreticalPER(20.0, 100) - 0.5735518487383953) < 1e-12
    assert abs(q64.calcTheoreticalSpectralEfficiency(20.0, 100) - 2.5586889075696284) < 1e-12
    assert modulators.BPSK(engine=engine).name == "BPSK"
    assert np.array_equal(modulators.BPSK(engine=engine).modulate(np.array([0, 1, 1])), [1.0, -1.0, -1.0])
    p = modulators.PSK(8, engine=engine)
    assert np.array_equal(p.demodulate(golden_ops["demod_psk8_rx"]), golden_ops["demod_psk8_dec"])
    o = modulators.OFDM(64, 16, 52, engine=engine)
    key = "ofdm_64_16_52"
    assert np.array_equal(o.get_used_subcarrier_indexes(), golden_ops[key + "_map"])
    assert relerr(o.modulate(golden_ops[key + "_x"]), golden_ops[key + "_tx"]) <= 1e-12
    assert relerr(o.demodulate(golden_ops[key + "_tx"]), golden_ops[key + "_back"]) <= 1e-11
    with pytest.raises(ValueError):
        modulators.OFDM(64, 65)
    b = mimo.Blast(golden_ops["blast_H"], engine=engine)
    assert np.array_equal(b.encode(golden_ops["blast_x"]), golden_ops["blast_enc"])
    assert relerr(b.decode(golden_ops["blast_y"]), golden_ops["blast_zf"]) <= 1e-10
    b.set_noise_var(float(golden_ops["blast_nv"]))
    assert relerr(b.decode(golden_ops["blast_y"]), golden_ops["blast_mmse"]) <= 1e-10
    with pytest.raises(ValueError):
        b.set_noise_var(-1.0)
    a = mimo.Alamouti(golden_ops["ala_H"], engine=engine)
    assert relerr(a.decode(golden_ops["ala_y"]), golden_ops["ala_dec"]) <= 1e-12
    with pytest.raises(ValueError):
        mimo.Alamouti(np.ones((2, 3), dtype=complex))
    m = mimo.MRT(golden_ops["mrt_h"], engine=engine)
    assert relerr(m.encode(golden_ops["mrt_x"]), golden_ops["mrt_enc"]) <= 1e-14
    # channels: Jakes bookkeeping (tests/channels_package_test.py:244-290) and the TDL property test
    rs = np.random.RandomState(5)
    jk = channels.JakesSampleGenerator(Fd=5, Ts=1e-3, L=16, RS=rs, engine=engine)
    jk.generate_more_samples(100)
    first = jk.get_samples().copy()
    assert first.shape == (100,) and abs(jk._current_time - 101e-3) < 1e-9
    jk.skip_samples_for_next_generation(50)
    assert abs(jk._current_time - 151e-3) < 1e-9
    tdl = channels.TdlChannel(channels.JakesSampleGenerator(Fd=5, Ts=3.255e-8, L=16, RS=rs, engine=engine),
                              channels.COST259_TUx, engine=engine)
    assert tdl.num_taps == 15 and tdl.num_taps_with_padding == 67
    # constructor known answers and errors (tests/channels_package_test.py:740-790)
    with pytest.raises(RuntimeError):
        tdl.get_last_impulse_response()          # nothing generated yet
    jk2 = channels.JakesSampleGenerator(Fd=5, Ts=3.255e-8, L=16, RS=rs, engine=engine)
    far = channels.TdlChannel(jk2, tap_powers_dB=channels.COST259_TUx.tap_powers_dB,
                              tap_delays=10 * channels.COST259_TUx.tap_delays)
    assert far.num_taps == 20 and far.num_taps_with_padding == 658 and far.channel_profile.Ts == 3.255e-8
    with pytest.raises(RuntimeError):             # Ts differs from the Jakes generator's
        channels.TdlChannel(jk2, channels.COST259_TUx, Ts=0.002)
    with pytest.raises(RuntimeError):             # profile discretised with another Ts
        channels.TdlChannel(jk2, channels.COST259_TUx.get_discretize_profile(0.002))
    ray = channels.TdlChannel(channels.RayleighSampleGenerator(), channels.COST259_TUx, Ts=3.255e-8)
    assert ray.num_taps == 15 and ray.num_taps_with_padding == 67
    x = rs.randn(300) + 1j * rs.randn(300)
    y = tdl.corrupt_data(x)
    ir = tdl.get_last_impulse_response()
    want = np.zeros(366, dtype=complex)
    for i, d in enumerate(ir.tap_indexes_sparse):
        want[d:d + 300] += ir.tap_values_sparse[i] * x
    assert y.shape == (366,) and relerr(y, want) <= 1e-13
    assert ir.tap_values.shape == (67, 300) and ir.get_freq_response(128).shape == (128, 300)
    su = channels.SuChannel(channels.JakesSampleGenerator(RS=rs, engine=engine), engine=engine)
    assert su.corrupt_data(x).shape == (300,) and su.num_taps == 1
    util.seed(7, 3)
    z = util.randn_c(4, 5, engine=engine)
    assert z.shape == (4, 5) and z.dtype == np.complex128
    assert util.count_bit_errors(np.array([[2, 3, 3, 0], [1, 3, 1, 2]]), np.array([[0, 3, 2, 0], [2, 0, 1, 2]]),
                                 engine=engine) == 6
    assert list(util.count_bit_errors(np.array([[2, 3, 3, 0], [1, 3, 1, 2]]), np.array([[0, 3, 2, 0], [2, 0, 1, 2]]),
                                      1, engine=engine)) == [2, 4]


def test_gmd_filters(engine, golden_ops):
    """GMDMimo: R (a function of the singular values only) equals the reference's util.misc.gmd output;
    W = P/sqrt(n) is unitary/sqrt(n); the zero-forcing filter on Q R inverts H W (H = Q R P^H)."""
    H = golden_ops["svd_H"]
    W, G, R = engine.gmd_filters(H[None], 0.0)
    assert relerr(R[0], golden_ops["gmd_R"]) <= 1e-11
    assert np.allclose(np.tril(R[0], -1), 0) and relerr(np.diag(R[0]), np.full(4, np.prod(golden_ops["svd_S"]) ** 0.25)) <= 1e-12
    assert relerr(W[0].conj().T @ W[0] * 4, np.eye(4)) <= 1e-12
    assert relerr(G[0] @ H @ W[0], np.eye(4)) <= 1e-9
    QR = H @ (W[0] * 2.0)                                   # = Q R
    assert relerr(np.abs(np.linalg.qr(QR)[1]), np.abs(R[0])) <= 1e-9      # same triangular factor up to phases
    from pyphysim_amd.mimo import GMDMimo
    m = GMDMimo(H, engine=engine)
    x = golden_ops["svd_x"]
    assert relerr(m.decode(H @ m.encode(x)), x) <= 1e-9 and relerr(golden_ops["gmd_dec"], x) <= 1e-9
    rs = np.random.RandomState(3)
    for n in (2, 3, 4):
        Hn = (rs.randn(5, n, n) + 1j * rs.randn(5, n, n)) / np.sqrt(2)
        Wn, Gn, Rn = engine.gmd_filters(Hn, 0.05)
        for b in range(5):
            S = np.linalg.svd(Hn[b])[1]
            assert relerr(np.diag(Rn[b]), np.full(n, np.prod(S) ** (1.0 / n))) <= 1e-11
            Heq = Hn[b] @ (Wn[b] * np.sqrt(n))
            want = np.sqrt(n) * np.linalg.solve(Heq.conj().T @ Heq + 0.05 * np.eye(n), Heq.conj().T)
            assert relerr(Gn[b], want) <= 1e-9             # Blast's MMSE on the equivalent channel


# ---- frequency-selective MIMO-OFDM (MIMO TDL channel + per-subcarrier MMSE) -------------
@pytest.mark.parametrize("dt", ["f64", "f32"])
def test_mimo_ofdm_tdl_chain_injected(engine, dt):
    for kw, reals in golden_cases("f1_mimo_ofdm_tdl"):
        g = reals[0]
        nt, nr, fft, cp = kw["nt"], kw["nr"], kw["fft_size"], kw["cp_size"]
        used, nsym, L = kw["num_used"] or fft, kw["n_ofdm_sym"], kw["L"]
        d_idx = g["delay_indexes"]
        S = len(d_idx)
        n = nsym * (fft + cp)
        t, _ = och.jakes_time_axis(kw["Ts"], kw["Ts"], n)
        P = S * nr * nt
        taps = engine.jakes_generate(g["phi"].reshape(L, P), g["psi"].reshape(L, P), kw["Fd"], kw["Ts"],
                                     float(t[1] - t[0]), n, tap_power=np.repeat(g["tap_powers_linear"], nr * nt),
                                     dtype=dt).reshape(S, nr, nt, n)
        assert relerr(taps, g["taps"]) <= (1e-9 if dt == "f64" else 3e-5)
        engine.set_constellation(g["table"], _lib.CONST_QAM)
        sym = engine.modulate(g["idx"], dtype=dt)
        X = engine.blast_encode(sym, nt, dtype=dt)[0]
        T = engine.ofdm_modulate(X, fft, cp, used, batch=nt, dtype=dt)
        assert relerr(T, g["T"]) <= TOL[dt]
        faded = engine.tdl_apply_mimo(T, taps, d_idx, dtype=dt)
        assert faded.shape == (nr, n + int(d_idx[-1]))
        R = engine.awgn_add(faded, g["noise"], float(g["noise_var"]), dtype=dt)
        Y = engine.ofdm_demodulate(np.ascontiguousarray(R[:, :n]), fft, cp, used, batch=nr, dtype=dt)
        assert relerr(Y, g["Y"]) <= (1e-9 if dt == "f64" else 1e-4)
        Hu = engine.tdl_mean_freq_response(taps, d_idx, nsym, fft, cp, used, dtype=dt).reshape(-1, nr, nt)
        assert relerr(Hu, g["Hu"]) <= (1e-9 if dt == "f64" else 1e-4)
        G, sk = engine.blast_filter(Hu, float(g["noise_var"]), dtype=dt)
        assert not sk.any()
        est = engine.blast_decode_per_subcarrier(G, Y, dtype=dt)
        assert relerr(est, g["est"]) <= (1e-8 if dt == "f64" else 5e-3)
        dec = engine.demodulate(est, dtype=dt)
        if dt == "f64":
            assert np.array_equal(dec, g["decisions"])
        else:
            assert np.count_nonzero(dec != g["decisions"]) <= 4


def test_mimo_tdl_mirror_classes(engine):
    """TdlMimoChannel / SuMimoChannel mirrors: explicit shifted sums (reference
    tests/channels_package_test.py:1162-1407 property) on the HIP path."""
    from pyphysim_amd import channels
    rs = np.random.RandomState(9)
    jk = channels.JakesSampleGenerator(Fd=30, Ts=1e-6, L=8, shape=(3, 2), RS=rs, engine=engine)
    tdl = channels.TdlMimoChannel(jk, tap_powers_dB=np.array([0.0, -3.0, -8.0]), tap_delays=np.array([0, 2, 7]) * 1e-6,
                                  engine=engine)
    assert (tdl.num_rx_antennas, tdl.num_tx_antennas, tdl.num_taps) == (3, 2, 3)
    x = rs.randn(2, 100) + 1j * rs.randn(2, 100)
    y = tdl.corrupt_data(x)
    ir = tdl.get_last_impulse_response()
    assert ir.tap_values_sparse.shape == (3, 3, 2, 100) and y.shape == (3, 107)
    assert relerr(y, och.tdl_apply_mimo(x, ir.tap_values_sparse, ir.tap_indexes_sparse)) <= 1e-13
    assert ir.get_freq_response(16).shape == (16, 3, 2, 100)
    with pytest.raises(RuntimeError):
        channels.TdlMimoChannel(channels.JakesSampleGenerator(RS=rs, engine=engine))
    su = channels.SuMimoChannel(2, channels.JakesSampleGenerator(RS=rs, engine=engine), engine=engine)
    assert su.corrupt_data(x).shape == (2, 100) and su.num_tx_antennas == 2
    # reverse link (reference tests/channels_package_test.py:1277-1407, :1526-1600): Nr-antenna signal in,
    # Nt-antenna signal out through the transposed taps
    with pytest.raises(TypeError):
        tdl.switched_direction = 1
    tdl.switched_direction = True
    xr = rs.randn(3, 64) + 1j * rs.randn(3, 64)
    yr = tdl.corrupt_data(xr)
    ir = tdl.get_last_impulse_response()
    want = np.zeros((2, 64 + 7), dtype=complex)
    for i, d in enumerate(ir.tap_indexes_sparse):
        for r in range(3):
            want[:, d:d + 64] += ir.tap_values_sparse[i, r, :, :] * xr[r]
    assert yr.shape == (2, 71) and relerr(yr, want) <= 1e-13
    # block-static shortcut in the same direction (fading.py:1254-1258)
    xf = rs.randn(3, 4 * 16) + 1j * rs.randn(3, 4 * 16)
    yf = tdl.corrupt_data_in_freq_domain(xf, 16)
    ir = tdl.get_last_impulse_response()
    fr = ir.get_freq_response(16)                                   # [16, 3, 2, 4]
    wantf = np.zeros((4 * 16, 2), dtype=complex)
    for b in range(4):
        for r in range(3):
            wantf[b * 16:(b + 1) * 16, :] += fr[:, r, :, b] * xf[r, b * 16:(b + 1) * 16, None]
    assert yf.shape == (2, 64) and relerr(yf, wantf.T) <= 1e-12


def test_large_and_degenerate_operator_inputs(engine, golden_ops):
    rs = np.random.RandomState(17)
    n = 1 << 22
    engine.set_constellation(golden_ops["qam64"], _lib.CONST_QAM)
    idx = rs.randint(0, 64, n)
    tx = engine.modulate(idx, dtype="f32")
    assert tx.shape == (n,) and np.array_equal(tx, golden_ops["qam64"][idx].astype(np.complex64))
    back = engine.demodulate(tx, dtype="f32")                      # noiseless: exact recovery, 4M symbols
    assert np.array_equal(back, idx)
    assert np.array_equal(engine.demodulate(tx, method=_lib.DEMOD_QAM_SLICER, dtype="f32"), idx)
    cnt, se, be = engine.count_errors(idx, np.roll(idx, 1), 6)
    assert int(se[0]) == int(np.sum(idx != np.roll(idx, 1))) and int(be[0]) == int(omodem.count_bit_errors(idx, np.roll(idx, 1)))
    # the largest table the operator kernels accept (setConstellation with an arbitrary point set)
    big = (rs.randn(1024) + 1j * rs.randn(1024))
    engine.set_constellation(big)
    pick = rs.randint(0, 1024, 5000)
    assert np.array_equal(engine.demodulate(engine.modulate(pick)), omodem.demodulate(big, big[pick]))
    with pytest.raises(_lib.McleError):
        engine.set_constellation(np.ones(3, dtype=complex))        # not a power of two
    # many short OFDM symbols in one launch, more rows than the grid cap
    x = rs.randn(300, 16 * 9) + 1j * rs.randn(300, 16 * 9)
    tx = engine.ofdm_modulate(x, 16, 2, 16, batch=300)
    assert relerr(tx[123], oofdm.modulate(x[123], 16, 2, 16)) <= 1e-13
    assert relerr(engine.ofdm_demodulate(tx, 16, 2, 16, batch=300), x) <= 1e-12
    # empty inputs flow through every operator
    e = np.zeros(0, dtype=complex)
    assert engine.awgn_add(e, e, 0.1).size == 0 and engine.cdiv(e, e).size == 0
    assert engine.randn_c(0, 1, 2).size == 0 and engine.rand_symbols(0, 4, 1, 2).size == 0
    # Philox positions beyond 2^32 samples stay addressable (block index = sample >> 1)
    z = engine.randn_c(8, 3, 4, first=(1 << 32) + 10, dtype="f64")
    assert relerr(z, P.cnormal(3, 4, 8, P.STREAM_NOISE, offset=(1 << 32) + 10)) <= 1e-13


def test_corrupt_data_in_freq_domain(engine, golden_ops):
    """TdlChannel.corrupt_data_in_freq_domain (fading.py:1126-1287) through the mirror classes: the mirror
    Jakes generator is handed the reference's phases, then block times, tap gains, per-block frequency
    response and the per-carrier products run on the GPU."""
    from pyphysim_amd import channels
    g = golden_ops

    def jakes_with(phi, psi, shape):
        jk = channels.JakesSampleGenerator(Fd=40.0, Ts=1e-5, L=8, shape=shape, RS=np.random.RandomState(0),
                                           engine=engine)
        return jk

    jk = jakes_with(g["fd_phi"], g["fd_psi"], None)
    td = channels.TdlChannel(jk, tap_powers_dB=np.array([0.0, -3.0, -7.0]), tap_delays=np.array([0, 2, 5]) * 1e-5,
                             engine=engine)
    jk._phi_l, jk._psi_l = g["fd_phi"].copy(), g["fd_psi"].copy()      # the reference's draws (after the re-draw)
    out = td.corrupt_data_in_freq_domain(g["fd_sig"], 32, g["fd_car"])
    assert relerr(td.get_last_impulse_response().tap_values_sparse, g["fd_taps"]) <= 1e-10
    assert relerr(out, g["fd_out"]) <= 1e-10
    assert relerr(och.corrupt_data_in_freq_domain(g["fd_sig"], g["fd_taps"], g["fd_delays"], 32, g["fd_car"]),
                  g["fd_out"]) <= 1e-13
    jm = jakes_with(g["fdm_phi"], g["fdm_psi"], (2, 3))
    tm = channels.TdlMimoChannel(jm, tap_powers_dB=np.array([0.0, -3.0, -7.0]), tap_delays=np.array([0, 2, 5]) * 1e-5,
                                 engine=engine)
    jm._phi_l, jm._psi_l = g["fdm_phi"].copy(), g["fdm_psi"].copy()
    outm = tm.corrupt_data_in_freq_domain(g["fdm_sig"], 16)
    assert outm.shape == (2, 32) and relerr(outm, g["fdm_out"]) <= 1e-10
    with pytest.raises(ValueError):
        td.corrupt_data_in_freq_domain(g["fd_sig"][:-1], 32, g["fd_car"])


@pytest.mark.parametrize("n", [1, 2, 3, 1001, 4096, 65537])
def test_elementwise_vector_paths_handle_any_length(engine, n):
    """f32 element-wise kernels move two samples per thread; odd lengths and the last sample go one by one."""
    rs = np.random.RandomState(n)
    table = omodem.qam_constellation(16)
    engine.set_constellation(table, _lib.CONST_QAM)
    idx = rs.randint(0, 16, n)
    a = (rs.randn(n) + 1j * rs.randn(n)).astype(np.complex64)
    b = (rs.randn(n) + 1j * rs.randn(n) + 3).astype(np.complex64)
    assert relerr(engine.modulate(idx, dtype="f32"), table[idx]) <= 1e-6
    assert relerr(engine.awgn_add(a, b, 0.25, dtype="f32"), a + 0.5 * b) <= 1e-6
    assert relerr(engine.cmul(a, b, dtype="f32"), a * b) <= 1e-6
    assert relerr(engine.cdiv(a, b, dtype="f32"), a / b) <= 1e-6
    rx = (table[idx] + 0.05 * a).astype(np.complex64)
    want = np.argmin(np.abs(rx[:, None] - table[None, :].astype(np.complex64)), axis=1)
    assert np.array_equal(engine.demodulate(rx, dtype="f32"), want)
    assert np.array_equal(engine.demodulate(rx, method=_lib.DEMOD_QAM_SLICER, dtype="f32"), want)


def test_ia_class_mirrors_run_the_reference_app(engine):
    """apps/ia/simulate_ia.py:94-245 written against the mirror classes (multiuser.MultiUserChannelMatrix +
    ia.*IASolver) with the reference's seeding: precoders, filters, iteration counts and decisions of the
    reference run (tests/golden/c5_ia.npz, f3_ia_iterative.npz)."""
    from pyphysim_amd import ia, multiuser
    from pyphysim_amd.modulators import QAM
    classes = {"closed_form": ia.ClosedFormIASolver, "alt_min": ia.AlternatingMinIASolver,
               "min_leakage": ia.MinLeakageIASolver, "max_sinr": ia.MaxSinrIASolver, "mmse": ia.MMSEIASolver}
    for name in ("c5_ia", "f3_ia_iterative"):
        for kw, reals in golden_cases(name):
            algo = kw.get("algo", "closed_form")
            for g in reals:
                seed = int(g["seed"])
                np.random.seed(seed)
                m = QAM(kw["M"], engine=engine)
                muc = multiuser.MultiUserChannelMatrix(engine=engine)
                muc.set_channel_seed(seed)
                muc.set_noise_seed(seed)
                solver = classes[algo](muc)
                if algo != "closed_form":
                    solver._rs = np.random.RandomState(seed)
                    solver._alt_min_rs = np.random.RandomState(seed)
                    solver.max_iterations = kw["max_iterations"]
                    solver.relative_factor = kw["relative_factor"]
                    solver.initialize_with = kw.get("initialize_with", "random")
                muc.randomize(kw["nr"], kw["nt"], kw["K"])
                muc.noise_var = float(g["noise_var"])
                assert relerr(muc.big_H, g["big_H"]) == 0.0
                solver.clear()
                solver.solve(kw["Ns"])
                if algo != "closed_form":
                    assert solver.runned_iterations == int(g["runned_iterations"])
                idx = np.random.randint(0, kw["M"], [3, kw["NSymbs"]])
                assert np.array_equal(idx, g["idx"])
                tx = np.split(m.modulate(idx), [1, 2])
                pre = [np.dot(f, x) for f, x in zip(solver.full_F, tx)]
                rx = muc.corrupt_data(pre)
                est = np.vstack([np.dot(u, y) for u, y in zip(solver.full_W_H, rx)])
                assert relerr(est, g["est"]) <= 1e-6
                assert np.array_equal(m.demodulate(est), g["decisions"])
                sinr = np.concatenate(list(solver.calc_SINR()))
                assert relerr(sinr, g["sinr"]) <= 1e-6
    muc = multiuser.MultiUserChannelMatrix(engine=engine)
    muc.randomize(3, 3, 3)
    muc.noise_var = 0.01
    assert ia.MaxSinrIASolver(muc).solve(1) >= 1        # 3x3 runs on the general-geometry kernel since round 2
    with pytest.raises(ValueError):
        ia.ClosedFormIASolver(muc).solve(1)             # the closed form stays K = 3, 2x2, one stream
    with pytest.raises(RuntimeError):
        ia.AlternatingMinIASolver(muc).initialize_with = "alt_min"
    with pytest.raises(ValueError):
        muc.init_from_channel_matrix(np.zeros((4, 4)), 2, 2, 3)


def test_post_processing_sinrs(engine, golden_ops):
    """calc_post_processing_(linear_)SINRs and every scheme's calc_linear_SINRs (reference mimo/mimo.py:33-118,
    311-345, 1147-1166) against the reference's own numbers."""
    from oracle import mimo as omimo
    from pyphysim_amd import mimo
    g = golden_ops
    got = mimo.calc_post_processing_linear_SINRs(g["psinr_H"], g["psinr_W"], g["psinr_G"], float(g["psinr_nv"]),
                                                 engine=engine)
    assert relerr(got, g["psinr_lin"]) <= 1e-12
    assert relerr(mimo.calc_post_processing_SINRs(g["psinr_H"], g["psinr_W"], g["psinr_G"], float(g["psinr_nv"]),
                                                  engine=engine), 10 * np.log10(g["psinr_lin"])) <= 1e-12
    batch = np.stack([g["psinr_H"], 2 * g["psinr_H"], g["psinr_H"].conj()])
    out = engine.post_processing_sinrs(batch, np.stack([g["psinr_W"]] * 3), np.stack([g["psinr_G"]] * 3), 0.07)
    for b in range(3):
        assert relerr(out[b], omimo.post_processing_linear_sinrs(batch[b], g["psinr_W"], g["psinr_G"], 0.07)) <= 1e-12
    # the schemes' own methods, the reference's quirk included: MimoBase.calc_linear_SINRs returns dB (mimo.py:325)
    Hq = g["psinr_Hq"]
    assert np.all(mimo.Blast(Hq, engine=engine).calc_linear_SINRs(0.0) > 250.0)         # zero forcing, no noise:
    assert np.all(g["psinr_blast_zf"] > 250.0)                                            # rounding residue only
    assert relerr(mimo.Blast(Hq, engine=engine).calc_linear_SINRs(0.05), g["psinr_blast_mmse"]) <= 1e-8
    assert relerr(mimo.SVDMimo(Hq, engine=engine).calc_linear_SINRs(0.05), g["psinr_svd"]) <= 1e-7
    assert relerr(mimo.GMDMimo(Hq, engine=engine).calc_linear_SINRs(0.05), g["psinr_gmd"]) <= 1e-7
    assert relerr(mimo.MRT(g["psinr_mrt_h"], engine=engine).calc_linear_SINRs(0.05), g["psinr_mrt"]) <= 1e-10
    assert relerr(mimo.MRC(g["psinr_mrt_h"], engine=engine).calc_linear_SINRs(0.05), g["psinr_mrc"]) <= 1e-8
    assert relerr(np.atleast_1d(mimo.Alamouti(g["psinr_ala_H"], engine=engine).calc_linear_SINRs(0.05)),
                  g["psinr_ala"]) <= 1e-12
    want_db = 10 * np.log10(np.abs(g["psinr_blast_mmse"]))
    got_db = mimo.Blast(Hq, engine=engine).calc_SINRs(0.05)
    assert relerr(got_db[np.isfinite(want_db)], want_db[np.isfinite(want_db)]) <= 1e-7


def test_generate_jakes_samples_function_and_concatenate(engine):
    """fading_generators.py:15-98 (the function form, unit-interval phases fed to cos() as they are) against its own
    closed form; TdlImpulseResponse.concatenate_samples (fading.py:655-698)."""
    from pyphysim_amd import channels
    rs = np.random.RandomState(12)
    L, shape, n, Fd, Ts, t0 = 8, (2, 3), 50, 30.0, 1e-3, 0.25
    phi, psi = rs.rand(L, 2, 3, 1), rs.rand(L, 2, 3, 1)
    t_next, h = channels.generate_jakes_samples(Fd, Ts, n, L, shape, t0, phi, psi, engine=engine)
    t = np.arange(t0, n * Ts + t0, Ts * 1.0000000001)
    want = math.sqrt(1.0 / L) * np.sum(np.exp(1j * (2 * np.pi * Fd * np.cos(phi) * t + psi)), axis=0)
    assert h.shape == (2, 3, n) and relerr(h, want) <= 1e-10 and abs(t_next - (t[-1] + Ts)) < 1e-15
    np.random.seed(5)
    _, h1 = channels.generate_jakes_samples(Fd, Ts, 10, engine=engine)          # draws its own phases, no shape
    np.random.seed(5)
    p1, p2 = np.random.rand(8, 1), np.random.rand(8, 1)
    t = np.arange(0, 10 * Ts, Ts * 1.0000000001)
    assert relerr(h1, math.sqrt(1.0 / 8) * np.sum(np.exp(1j * (2 * np.pi * Fd * np.cos(p1) * t + p2)), axis=0)) <= 1e-10
    prof = channels.TdlChannelProfile(np.array([0.0, -3.0]), np.array([0.0, 1e-6])).get_discretize_profile(1e-6)
    a = channels.TdlImpulseResponse(rs.randn(2, 4) + 0j, prof)
    b = channels.TdlImpulseResponse(rs.randn(2, 3) + 0j, prof)
    c = channels.TdlImpulseResponse.concatenate_samples([a, b])
    assert c.num_samples == 7 and np.array_equal(c.tap_values_sparse[:, 4:], b.tap_values_sparse)
    assert channels.TdlImpulseResponse.concatenate_samples([a]) is a
    with pytest.raises(ValueError):
        channels.TdlImpulseResponse.concatenate_samples([])


def test_ofdm_1024_complex64_matrix_core_kernels(engine):
    """kernels_ofdm_mfma.hip: batches, a zero-padded last symbol, symbol counts that are not a multiple of the four
    symbols of a pass, partial bands, odd CP -- against the oracle, and against the radix-4 kernels it replaces."""
    import os
    rs = np.random.RandomState(21)
    for batch, n_sym_in, cp, used in ((1, 1, 16, 1024), (3, 5, 16, 1024), (2, 7, 73, 600), (5, 2, 0, 1022), (1, 9, 1024, 2)):
        n_in = n_sym_in * used - (used // 3 if n_sym_in > 1 else 0)          # the last symbol is zero padded
        x = (rs.randn(batch, n_in) + 1j * rs.randn(batch, n_in)).astype(np.complex64)
        tx = engine.ofdm_modulate(x, 1024, cp, used, batch=batch, dtype="f32")
        want = np.stack([oofdm.modulate(x[b].astype(complex), 1024, cp, used) for b in range(batch)])
        assert tx.shape == want.shape and relerr(tx, want) <= 2e-6, (batch, n_sym_in, cp, used, relerr(tx, want))
        back = engine.ofdm_demodulate(want, 1024, cp, used, batch=batch, dtype="f32")
        wantb = np.stack([oofdm.demodulate(want[b], 1024, cp, used) for b in range(batch)])
        assert back.shape == wantb.shape and relerr(back, wantb) <= 2e-6
        with engine.options(no_mfma=1):
            tx_v = engine.ofdm_modulate(x, 1024, cp, used, batch=batch, dtype="f32")
            back_v = engine.ofdm_demodulate(want, 1024, cp, used, batch=batch, dtype="f32")
        assert relerr(tx, tx_v) <= 2e-6 and relerr(back, back_v) <= 2e-6


@pytest.mark.parametrize("mod,M", [("qam", 16), ("qam", 64), ("qam", 256), ("psk", 8), ("psk", 16), ("qpsk", 4)])
def test_complex128_pruned_search_equals_the_sweep_on_ties_and_edges(engine, mod, M):
    """The complex128 demodulator runs the candidate-grid search too (round 2): the nearest constellation point on noise,
    on exact midpoints between constellation points (ties), on far outliers and on points a rounding step either side of
    the grid's cell edges."""
    table = chains.constellation(mod, M)
    kind = _lib.CONST_QAM if mod == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(table, kind)
    rs = np.random.RandomState(M)
    pts = [0.7 * (rs.randn(20000) + 1j * rs.randn(20000))]
    mid = 0.5 * (table[:, None] + table[None, :]).ravel()
    pts += [mid, mid * (1 + 1e-16), mid + 1e-9, 30.0 * (rs.randn(500) + 1j * rs.randn(500))]
    edges = np.linspace(-2.0, 2.0, 129)
    ex, ey = np.meshgrid(edges, edges[::7])
    pts += [(ex + 1j * ey).ravel(), np.nextafter(ex, 9.0).ravel() + 1j * np.nextafter(ey, -9.0).ravel()]
    r = np.concatenate(pts)
    d = np.abs(table[None, :] - r[:, None]) ** 2
    want = np.argmin(d, axis=1)
    got = engine.demodulate(r, dtype="f64")
    # away from exact ties the decision is the sweep's; ON a tie (the midpoints) which of the equidistant points wins
    # depends on the last bit of the metric -- hypot in NumPy, dx*dx + dy*dy (possibly fused) on the device -- so there
    # the chosen point only has to be one of the nearest
    rows = np.arange(r.size)
    gap = d[rows, got] - d[rows, want]
    assert np.all(gap <= 1e-12 * (1.0 + d[rows, want]))
    clear = np.partition(d, 1, axis=1)
    clear = (clear[:, 1] - clear[:, 0]) > 1e-9 * (1.0 + clear[:, 0])
    assert np.array_equal(got[clear], want[clear]) and clear.sum() > 20000
