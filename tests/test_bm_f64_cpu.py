"""CPU: the complex128 Box-Muller of csrc/bm_f64.hpp (table + short polynomial forms of -ln u, sqrt, sincos), compiled
for the host from the SAME header the device kernels include, against NumPy -- the functions oracle/philox.py::cnormal
is written with -- word by word.  The GPU parity tests then hold the pipelines' error counts equal to the oracle's."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bm(tmp_path_factory):
    out = tmp_path_factory.mktemp("bm") / "libbm_host.so"
    src = os.path.join(REPO, "tests", "host", "bm_f64_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    P = ctypes.c_void_p
    lib.bm_neg_log_batch.argtypes = [P, P, ctypes.c_size_t]
    lib.bm_sqrt_batch.argtypes = [P, P, ctypes.c_size_t]
    lib.bm_sincos_batch.argtypes = [P, P, P, ctypes.c_size_t]
    lib.bm_sincos_rad_batch.argtypes = [P, P, P, ctypes.c_size_t]
    return lib


def _words(n, seed):
    rs = np.random.RandomState(seed)
    w = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    edge = np.array([0, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1, 2 ** 31 + 1, 2 ** 23, 2 ** 23 - 1, 2 ** 24,
                     2 ** 32 - 2 ** 23, 2 ** 32 - 2 ** 23 - 1, 2 ** 30, 3 * 2 ** 30], dtype=np.uint64).astype(np.uint32)
    w[:edge.size] = edge
    # every table node boundary of both functions
    k = np.arange(512, dtype=np.uint64)
    w[100:100 + 512] = ((k << np.uint64(23)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    w[700:700 + 512] = (((k << np.uint64(23)) - np.uint64(1)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    # the words nearest u = 1 (where ln u cancels) and u = 0
    w[2000:12000] = (2 ** 32 - 1 - np.arange(10000) * 53).astype(np.uint32)
    w[12000:14000] = np.arange(2000, dtype=np.uint32)
    return np.ascontiguousarray(w)


def test_neg_log_matches_numpy(bm):
    x0 = _words(1_000_000, 5)
    got = np.empty(x0.size)
    bm.bm_neg_log_batch(x0.ctypes.data, got.ctypes.data, x0.size)
    want = -np.log((x0.astype(np.float64) + 0.5) * 2.0 ** -32)
    rel = np.abs(got - want) / want
    assert rel.max() <= 4.5e-16, rel.max()                         # <= 2 ulp of each other; each is <= ~2.3e-16 from the true value
    # what the sample sees: sqrt(-ln u) within 1.5 units in the last place (each side rounds its own square root)
    assert np.max(np.abs(np.sqrt(got) - np.sqrt(want)) / np.sqrt(want)) <= 3.4e-16
    assert np.all(got > 0) and np.isfinite(got).all()
    # the largest u: -ln(1 - 2^-33) = 2^-33 (1 + 2^-34 + ...) to full relative accuracy
    one = np.array([2 ** 32 - 1], dtype=np.uint32)
    g1 = np.empty(1)
    bm.bm_neg_log_batch(one.ctypes.data, g1.ctypes.data, 1)
    assert abs(g1[0] / 2.0 ** -33 - (1 + 2.0 ** -34)) <= 3e-16


def test_sqrt_is_correctly_rounded_on_the_range(bm):
    rs = np.random.RandomState(9)
    a = np.concatenate([np.exp(rs.uniform(np.log(2e-10), np.log(23.0), 500_000)), [2.0 ** -33, 22.2, 1.0, 4.0, 2.0]])
    got = np.empty(a.size)
    bm.bm_sqrt_batch(a.ctypes.data, got.ctypes.data, a.size)
    assert np.array_equal(got, np.sqrt(a))


def test_sincos_matches_numpy_on_the_double_angle(bm):
    x1 = _words(1_000_000, 6)
    c, s = np.empty(x1.size), np.empty(x1.size)
    bm.bm_sincos_batch(x1.ctypes.data, c.ctypes.data, s.ctypes.data, x1.size)
    ang = 2.0 * np.pi * (x1.astype(np.float64) * 2.0 ** -32)       # oracle/philox.py cnormal
    assert np.max(np.abs(c - np.cos(ang))) <= 2.3e-16
    assert np.max(np.abs(s - np.sin(ang))) <= 2.3e-16
    assert np.max(np.abs(c * c + s * s - 1.0)) <= 5e-16
    # exact axis values
    ax = np.array([0, 2 ** 30, 2 ** 31, 3 * 2 ** 30], dtype=np.uint32)
    ca, sa = np.empty(4), np.empty(4)
    bm.bm_sincos_batch(ax.ctypes.data, ca.ctypes.data, sa.ctypes.data, 4)
    assert np.allclose(ca, [1, 0, -1, 0], atol=2.5e-16) and np.allclose(sa, [0, 1, 0, -1], atol=2.5e-16)


def test_general_argument_sincos_matches_numpy(bm):
    """bm_sincos_rad: the Jakes ray phases of the complex128 kernels, x = 2 pi Fd cos(phi) t + psi up to ~1e5 rad."""
    rs = np.random.RandomState(7)
    x = np.concatenate([rs.uniform(-1e5, 1e5, 600_000), rs.uniform(-10, 10, 200_000), rs.uniform(-2 ** 24, 2 ** 24, 100_000),
                        2 * np.pi * np.arange(-300, 300) / 128.0, [0.0, -0.0, 1e-300, 6.283185307179586, 62831.853, -3.1415926]])
    # Jakes phases proper: w t + psi, w = 2 pi 100 cos(phi), t up to 100 s
    phi, psi = rs.uniform(0, 2 * np.pi, 100_000), rs.uniform(0, 2 * np.pi, 100_000)
    x = np.concatenate([x, (2 * np.pi * 100.0 * np.cos(phi)) * rs.uniform(0, 100, 100_000) + psi])
    x = np.ascontiguousarray(x)
    c, s = np.empty(x.size), np.empty(x.size)
    bm.bm_sincos_rad_batch(x.ctypes.data, c.ctypes.data, s.ctypes.data, x.size)
    LD = np.longdouble
    if np.finfo(LD).nmant >= 63:          # against extended precision where the host has it (its own reduction is exact to 1e-19 x)
        small = np.abs(x) <= 1e5
        ref_c, ref_s = np.cos(x[small].astype(LD)), np.sin(x[small].astype(LD))
        assert np.max(np.abs(c[small].astype(LD) - ref_c)) <= 1.8e-16
        assert np.max(np.abs(s[small].astype(LD) - ref_s)) <= 1.8e-16
    assert np.max(np.abs(c - np.cos(x))) <= 2.3e-16 and np.max(np.abs(s - np.sin(x))) <= 2.3e-16
    assert np.max(np.abs(c * c + s * s - 1.0)) <= 5e-16


def test_tables_regenerate_identically(tmp_path):
    """bm_tables.hpp is the output of scripts/gen_bm_tables.py (x87 extended precision rounded once)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_bm_tables", os.path.join(REPO, "scripts", "gen_bm_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    inv, lnc, theta, ct, st, step = gen.tables()
    text = open(os.path.join(REPO, "pyphysim_amd", "csrc", "bm_tables.hpp")).read()
    for v in (inv[1], lnc[37], theta[100], ct[33], st[127], ct[128], st[128]):
        assert float(v).hex() in text
