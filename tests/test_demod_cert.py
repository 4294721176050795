"""The margin certificate of the square-QAM min-distance decision (csrc/modem.hpp demod_qam_cert).

CPU: a NumPy statement of the same arithmetic (level coordinate, clamp, rint, |f| <= 1/2 - eps, packed Gray decode) against
the exhaustive |c - r| argmin of the reference (modulators/fundamental.py:241-246) on random points and on points placed at
decision boundaries +- tiny offsets: wherever the certificate says `sure` its label IS the argmin; the uncertified share is
what the table search still serves.  GPU: the demodulate operator with the certificate (default), with it switched off
(option demod_nocert: candidate grid) and NumPy's argmin agree on every point, boundary points included."""
import numpy as np
import pytest

from pyphysim_amd import _lib
from pyphysim_amd.modulators import constellation


def cert_model(r, M, dtype):
    """(labels, sure) as demod_qam_cert computes them (same operations, same order, in `dtype`)."""
    T = np.float64 if dtype == "f64" else np.float32
    bits = int(np.log2(M))
    hb = bits // 2
    L = 1 << hb
    scale = T(np.sqrt((M - 1) * 2.0 / 3.0))
    lim = T(0.5 - 2.0 ** -30) if dtype == "f64" else T(0.5 - 2.0 ** -15)
    lm1, hs, hl = T(L - 1), scale * T(0.5), T(L - 1) * T(0.5)
    x, y = r.real.astype(T), r.imag.astype(T)
    tj = np.minimum(np.maximum(x * hs + hl, T(0)), lm1)
    ti = np.minimum(np.maximum(hl - y * hs, T(0)), lm1)
    kj, ki = np.rint(tj), np.rint(ti)
    sure = (np.abs(tj - kj) <= lim) & (np.abs(ti - ki) <= lim)
    if dtype == "f32":                                               # round 5: complex64 certifies nothing far outside the constellation
        rmax = (T(16.0) + hl) / hs
        sure &= (np.abs(x) <= rmax) & (np.abs(y) <= rmax)
    v = (ki.astype(np.uint32) << 8) | kj.astype(np.uint32)
    v ^= (v >> 4) & 0x0F0F
    v ^= (v >> 2) & 0x3F3F
    v ^= (v >> 1) & 0x7F7F
    return (((v >> 8) << hb) | (v & 0xFF)).astype(np.int64), sure


def adversarial_points(M, rng, n=20000):
    """random points over and beyond the constellation + points on decision boundaries +- {0, 1e-15 .. 1e-3} spacings."""
    tab = constellation("qam", M)
    L = int(np.sqrt(M))
    spacing = 2.0 / np.sqrt((M - 1) * 2.0 / 3.0)
    amp = np.abs(tab.real).max()
    pts = [(rng.uniform(-1.6, 1.6, n) + 1j * rng.uniform(-1.6, 1.6, n)) * amp,
           tab[rng.integers(0, M, n)] + 0.35 * spacing * (rng.standard_normal(n) + 1j * rng.standard_normal(n))]
    bounds = (np.arange(L - 1) - (L - 2) / 2.0) * spacing            # the L - 1 decision boundaries of an axis
    for off in (0.0, 1e-15, 1e-13, 1e-11, 2.0 ** -31, 2.0 ** -29, 1e-8, 1e-6, 2.0 ** -13, 2.0 ** -11, 1e-3):
        for sgn in (-1.0, 1.0):
            b = rng.choice(bounds, 500) + sgn * off * spacing
            other = rng.uniform(-1.3, 1.3, 500) * amp
            pts.append(b + 1j * other)
            pts.append(other + 1j * b)
            pts.append(b + 1j * (rng.choice(bounds, 500) - sgn * off * spacing))    # corners of four regions
    return tab, np.concatenate(pts)


@pytest.mark.parametrize("M", [4, 16, 64, 256])
def test_certificate_model_equals_the_argmin_where_sure(M):
    rng = np.random.default_rng(M)
    tab, r = adversarial_points(M, rng)
    want = np.argmin(np.abs(tab[None, :] - r[:, None]), axis=1)
    lab, sure = cert_model(r, M, "f64")
    assert np.array_equal(lab[sure], want[sure])
    assert sure.mean() > 0.6                                  # the adversarial set is half boundary points
    # the random half alone: practically everything certified
    lab0, sure0 = cert_model(r[:40000], M, "f64")
    assert sure0.mean() > 1.0 - 1e-4
    # complex64: same statement against the argmin of the float-rounded point in exact arithmetic
    r32 = r.astype(np.complex64)
    want32 = np.argmin(np.abs(tab[None, :] - r32.astype(np.complex128)[:, None]), axis=1)
    lab32, sure32 = cert_model(r32, M, "f32")
    assert np.array_equal(lab32[sure32], want32[sure32])
    assert sure32[:40000].mean() > 1.0 - 5e-4 * np.sqrt(M)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [4, 16, 64, 256])
def test_demodulate_with_and_without_the_certificate(engine, M):
    rng = np.random.default_rng(100 + M)
    tab, r = adversarial_points(M, rng)
    engine.set_constellation(tab, _lib.CONST_QAM)
    d = np.abs(tab[None, :] - r[:, None])
    want = np.argmin(d, axis=1)
    # a point whose two best |c - r| agree to rounding has no decision to be identical to (hypot vs squared metric)
    part = np.partition(d, 1, axis=1)
    clear = (part[:, 1] - part[:, 0]) > 1e-13
    got = engine.demodulate(r, dtype="f64")
    with engine.options(demod_nocert=1):
        grid = engine.demodulate(r, dtype="f64")
    assert np.array_equal(got, grid), np.flatnonzero(got != grid)[:10]      # certificate == table search, boundary points included
    assert np.array_equal(got[clear], want[clear]), np.flatnonzero(clear & (got != want))[:10]
    assert clear.mean() > 0.8, clear.mean()              # (the boundary points at offsets <= 1e-13 are not)
    # complex64: certificate and table search agree wherever the float metric has a clear winner
    r32 = r.astype(np.complex64)
    got32 = engine.demodulate(r32, dtype="f32")
    with engine.options(demod_nocert=1):
        grid32 = engine.demodulate(r32, dtype="f32")
    d32 = np.abs(tab[None, :] - r32.astype(np.complex128)[:, None])
    p32 = np.partition(d32, 1, axis=1)
    clear32 = (p32[:, 1] - p32[:, 0]) > 1e-5
    assert np.array_equal(got32[clear32], grid32[clear32])
    assert np.array_equal(got32[clear32], np.argmin(d32, axis=1)[clear32])


def quad_cert_model(r, tab, dtype):
    """csrc/modem.hpp::demod_quad_cert restated: four points (+-a, +-b), one per quadrant -> (label, sure)."""
    a, b = abs(tab[0].real), abs(tab[0].imag)
    lut = np.zeros(4, dtype=np.int64)
    for m, c in enumerate(tab):
        lut[(1 if c.real < 0 else 0) | (2 if c.imag < 0 else 0)] = m
    lo = min(a, b) * (2.0 ** -30 if dtype == np.float64 else 2.0 ** -15)
    hi = max(a, b) * 256.0
    ax, ay = np.abs(r.real), np.abs(r.imag)
    sure = (ax >= lo) & (ay >= lo) & (ax <= hi) & (ay <= hi)
    return lut[(r.real < 0).astype(int) | ((r.imag < 0).astype(int) << 1)], sure


def _quad_points(rng, tab, n=20000):
    a = abs(tab[0].real)
    r = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1.2 * a
    k = n // 4                                  # on and next to the axes, the origin, far out (deep-fade equaliser outputs)
    off = rng.choice([0.0, 1e-16, 1e-13, 1e-10, 2.0 ** -31, 2.0 ** -29, 1e-6, 1e-3], size=k) * a * rng.choice([-1, 1], size=k)
    r[:k // 2] = off[:k // 2] + 1j * r[:k // 2].imag
    r[k // 2:k] = r[k // 2:k].real + 1j * off[k // 2:]
    r[k:k + 8] = [0, 1e-300, a * 1e-20 * (1 + 1j), 300 * a, -300j * a, (255 + 255j) * a, (257 - 1j) * a, -(1e6 + 1e6j) * a]
    return r


def test_quadrant_certificate_model_equals_the_argmin_where_sure():
    rng = np.random.default_rng(44)
    for tab in (constellation("qpsk", 4), constellation("psk", 4), constellation("qam", 4)[[2, 0, 3, 1]],
                np.array([0.3 + 1.1j, -0.3 + 1.1j, 0.3 - 1.1j, -0.3 - 1.1j])):           # (a rectangle: a != b)
        tab = np.asarray(tab, dtype=np.complex128)
        if min(abs(tab[0].real), abs(tab[0].imag)) < 1e-9:
            continue                             # (PSK(4) without offset sits ON the axes: no quadrant certificate for it)
        r = _quad_points(rng, tab)
        lab, sure = quad_cert_model(r, tab, np.float64)
        want = np.argmin(np.abs(tab[None, :] - r[:, None]), axis=1)
        assert np.array_equal(lab[sure], want[sure])
        assert 0.7 < sure.mean() < 1.0


def axis4_cert_model(r, tab):
    """csrc/modem.hpp::demod_axis4_cert restated (complex128): four points (+-a, 0), (0, +-a) -> (label, sure)."""
    a = float(np.max(np.abs(np.concatenate([tab.real, tab.imag]))))
    lut = np.zeros(4, dtype=np.int64)
    for m, c in enumerate(tab):
        lut[(1 if c.real - c.imag < 0 else 0) | (2 if c.real + c.imag < 0 else 0)] = m
    u, v = r.real - r.imag, r.real + r.imag
    lo, hi = a * 2.0 ** -30, a * 256.0
    sure = (np.abs(u) >= lo) & (np.abs(v) >= lo) & (np.abs(r.real) <= hi) & (np.abs(r.imag) <= hi)
    return lut[(u < 0).astype(int) | ((v < 0).astype(int) << 1)], sure


def test_axis_certificate_model_equals_the_argmin_where_sure():
    """The reference's PSK(4) = exp(j 2 pi m / 4) sits ON the axes (no quadrant certificate): the decision regions are bounded by the
    diagonals.  Points on and next to the diagonals (offsets from 1e-16 to 1e-3 of a), the origin, far out."""
    rng = np.random.default_rng(46)
    for tab in (constellation("psk", 4), constellation("psk", 4)[[3, 1, 0, 2]], 2.5 * np.asarray(constellation("psk", 4))):
        tab = np.asarray(tab, dtype=np.complex128)
        a = float(np.max(np.abs(tab.real)))
        n = 20000
        r = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 1.2 * a
        k = n // 4
        off = rng.choice([0.0, 1e-16, 1e-13, 1e-10, 2.0 ** -31, 2.0 ** -29, 1e-6, 1e-3], size=k) * a * rng.choice([-1, 1], size=k)
        r[:k // 2] = r[:k // 2].real * (1 + 1j) + off[:k // 2]                # next to the diagonal re = im
        r[k // 2:k] = r[k // 2:k].real * (1 - 1j) + 1j * off[k // 2:]         # next to re = -im
        r[k:k + 8] = [0, 1e-300, a * 1e-20 * (1 + 1j), 300 * a, -300j * a, (255 + 200j) * a, (257 - 1j) * a, -(1e6 + 2e6j) * a]
        lab, sure = axis4_cert_model(r, tab)
        want = np.argmin(np.abs(tab[None, :] - r[:, None]), axis=1)
        assert np.array_equal(lab[sure], want[sure])
        assert 0.7 < sure.mean() < 1.0
        # ... and also with the metric the kernels sweep with, |c - r|^2
        want2 = np.argmin((tab.real[None, :] - r.real[:, None]) ** 2 + (tab.imag[None, :] - r.imag[:, None]) ** 2, axis=1)
        assert np.array_equal(lab[sure], want2[sure])


@pytest.mark.gpu
def test_qpsk_demodulate_with_and_without_the_quadrant_certificate(engine):
    rng = np.random.default_rng(45)
    for mod, kind in (("qpsk", _lib.CONST_GENERIC), ("qam", _lib.CONST_QAM)):
        tab = np.asarray(constellation(mod, 4), dtype=np.complex128)
        r = _quad_points(rng, tab)
        engine.set_constellation(tab, kind)
        d = np.abs(tab[None, :] - r[:, None])
        want = np.argmin(d, axis=1)
        part = np.partition(d, 1, axis=1)
        clear = (part[:, 1] - part[:, 0]) > 1e-13 * np.maximum(1.0, np.abs(r))
        got = engine.demodulate(r, dtype="f64")
        with engine.options(demod_nocert=1):
            grid = engine.demodulate(r, dtype="f64")
        assert np.array_equal(got, grid), np.flatnonzero(got != grid)[:10]      # certificate == table search, axes included
        assert np.array_equal(got[clear], want[clear]), np.flatnonzero(clear & (got != want))[:10]
        r32 = r.astype(np.complex64)
        got32 = engine.demodulate(r32, dtype="f32")
        with engine.options(demod_nocert=1):
            grid32 = engine.demodulate(r32, dtype="f32")
        d32 = np.abs(tab[None, :] - r32.astype(np.complex128)[:, None])
        p32 = np.partition(d32, 1, axis=1)
        clear32 = (p32[:, 1] - p32[:, 0]) > 1e-5 * np.maximum(1.0, np.abs(r32))
        assert np.array_equal(got32[clear32], grid32[clear32])
        assert np.array_equal(got32[clear32], np.argmin(d32, axis=1)[clear32])


# ---- M-PSK beyond four points: the sector certificate (csrc/modem.hpp demod_psk_cert, round 5) -------------------------------
def psk_cert_model(r, tab, dtype):
    """demod_psk_cert restated in NumPy (same operations, same order, in `dtype`) -> (label, sure)."""
    T = np.float64 if dtype == "f64" else np.float32
    M = len(tab)
    rad = abs(tab[0])
    a0 = np.arctan2(tab[0].imag, tab[0].real)
    phi0 = np.fmod(a0, 2 * np.pi / M)
    if phi0 < 0:
        phi0 += 2 * np.pi / M
    if 2 * np.pi / M - phi0 < 1e-9:
        phi0 = 0.0
    lut = np.zeros(M, dtype=np.int64)
    for m, c in enumerate(tab):
        lut[int(np.rint((np.arctan2(c.imag, c.real) - phi0) / (2 * np.pi / M))) % M] = m
    rc, rs = T(np.cos(phi0)), T(-np.sin(phi0))
    x, y = r.real.astype(T), r.imag.astype(T)
    ux, uy = x * rc - y * rs, x * rs + y * rc
    ax, ay = np.abs(ux), np.abs(uy)
    sw = ay > ax
    hi, lo = np.where(sw, ay, ax), np.where(sw, ax, ay)
    eps = T(2.0 ** -28) if dtype == "f64" else T(2.0 ** -12)
    tol = eps * hi
    p = np.zeros(r.shape, dtype=np.int64)
    ok = (hi >= T(rad * (2.0 ** -8 if dtype == "f64" else 0.125))) & (hi <= T(rad * (2.0 ** 8 if dtype == "f64" else 8.0)))
    for j in range(M // 8):
        th = (2 * j + 1) * np.pi / M
        d = lo * T(np.cos(th)) - hi * T(np.sin(th))
        p += d > 0
        ok &= np.abs(d) >= tol
    k = np.where(sw, M // 4 - p, p)
    k = np.where(ux < 0, M // 2 - k, k)
    k = np.where(uy < 0, -k, k) & (M - 1)
    return lut[k], ok


def _psk_points(rng, tab, n=30000):
    """random points, points next to every sector boundary (angular offsets 0 .. 1e-3 rad), on the fold lines (axes, diagonals),
    near the origin and far out"""
    M = len(tab)
    rad = abs(tab[0])
    r = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rad
    k = n // 3
    ang0 = np.angle(tab[0])
    bnd = ang0 + (2 * rng.integers(0, M, k) + 1) * np.pi / M
    off = rng.choice([0.0, 1e-16, 1e-13, 1e-10, 2.0 ** -30, 2.0 ** -27, 1e-6, 2.0 ** -13, 2.0 ** -10, 1e-3], size=k) * rng.choice([-1, 1], size=k)
    r[:k] = rad * rng.uniform(0.05, 3.0, k) * np.exp(1j * (bnd + off))
    fold = ang0 + rng.integers(0, 8, k // 4) * np.pi / 4 + rng.choice([0.0, 1e-16, -1e-16, 1e-9, -1e-9], size=k // 4)
    r[k:k + k // 4] = rad * rng.uniform(0.2, 2.0, k // 4) * np.exp(1j * fold)
    r[-6:] = [0, 1e-300, rad * 1e-12 * (1 + 1j), 1e4 * rad, -1e4j * rad, (300 + 300j) * rad]
    return r


@pytest.mark.parametrize("M,offset", [(8, 0.0), (8, np.pi / 8), (16, 0.0), (16, 0.3)])
def test_psk_sector_certificate_model_equals_the_argmin_where_sure(M, offset):
    """Wherever the sector certificate says `sure` its label IS the exhaustive |c - r| argmin of the reference
    (modulators/fundamental.py:241-246) -- sector boundaries at tiny angular offsets, fold lines and extreme magnitudes included."""
    from oracle import modem as omodem
    tab = np.asarray(omodem.psk_table(M, offset) if hasattr(omodem, "psk_table") else constellation("psk", M) * np.exp(1j * offset),
                     dtype=np.complex128)
    rng = np.random.default_rng(1000 + M)
    r = _psk_points(rng, tab)
    want = np.argmin(np.abs(tab[None, :] - r[:, None]), axis=1)
    lab, sure = psk_cert_model(r, tab, "f64")
    assert np.array_equal(lab[sure], want[sure]), np.flatnonzero(sure & (lab != want))[:10]
    assert 0.6 < sure.mean() < 1.0 and sure[20000:].mean() > 0.97          # the random third: nearly everything certified
    r32 = r.astype(np.complex64)
    want32 = np.argmin(np.abs(tab[None, :] - r32.astype(np.complex128)[:, None]), axis=1)
    lab32, sure32 = psk_cert_model(r32, tab, "f32")
    d32 = np.abs(tab[None, :] - r32.astype(np.complex128)[:, None])
    p32 = np.partition(d32, 1, axis=1)
    assert np.array_equal(lab32[sure32], want32[sure32]), np.flatnonzero(sure32 & (lab32 != want32))[:10]
    assert sure32[20000:].mean() > 0.85


@pytest.mark.gpu
@pytest.mark.parametrize("M,offset", [(8, 0.0), (8, np.pi / 8), (16, 0.0), (16, 0.3)])
def test_psk_demodulate_with_and_without_the_sector_certificate(engine, M, offset):
    tab = np.asarray(constellation("psk", M) * np.exp(1j * offset), dtype=np.complex128)
    rng = np.random.default_rng(2000 + M)
    r = _psk_points(rng, tab)
    engine.set_constellation(tab, _lib.CONST_GENERIC)
    d = np.abs(tab[None, :] - r[:, None])
    want = np.argmin(d, axis=1)
    part = np.partition(d, 1, axis=1)
    clear = (part[:, 1] - part[:, 0]) > 1e-13
    got = engine.demodulate(r, dtype="f64")
    with engine.options(demod_nocert=1):
        grid = engine.demodulate(r, dtype="f64")
    assert np.array_equal(got, grid), np.flatnonzero(got != grid)[:10]      # certificate == table search on every point
    assert np.array_equal(got[clear], want[clear])
    r32 = r.astype(np.complex64)
    got32 = engine.demodulate(r32, dtype="f32")
    with engine.options(demod_nocert=1):
        grid32 = engine.demodulate(r32, dtype="f32")
    d32 = np.abs(tab[None, :] - r32.astype(np.complex128)[:, None])
    p32 = np.partition(d32, 1, axis=1)
    clear32 = (p32[:, 1] - p32[:, 0]) > 1e-5 * np.maximum(1.0, np.abs(r32))
    assert np.array_equal(got32[clear32], grid32[clear32])
    assert np.array_equal(got32[clear32], np.argmin(d32, axis=1)[clear32])
