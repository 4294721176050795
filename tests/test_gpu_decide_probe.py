"""GPU: the decision forms of csrc/walk_f64.hpp (walk_decide<double, DEC, 4>: the packed walks of configs 5 / f6, config 2 and config
4's complex128 family decide through it) on points CHOSEN BY THE TEST, against numpy.argmin over the constellation (the reference's
demodulate, modulators/fundamental.py:241-246: first minimum of |r - c|^2).

The pipelines draw their estimates, so the branch that serves a symbol the margin certificate does not vouch for -- within 2^-30 of
a decision boundary, beyond the certificate's range -- runs about once in 1e8 symbols there and the pipeline tests cannot hold it
to anything.  Here every probe point can be such a symbol: 2^-42 ... 2^-20 of the spacing off the boundaries on either side, exact
ties on the symmetry axes (first index wins), far outside the constellation, and ordinary random points.  tests/gpu_src/decide_probe.hip
is compiled at test time (hipcc, ~3 s) against the headers the library is built from; nothing in the product refers to it."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import chains
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WDEC = {0: "generic", 1: "slicer", 2: "qam_cert", 3: "quad_cert", 4: "axis4_cert"}


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    out = tmp_path_factory.mktemp("probe") / "libdecide_probe.so"
    src = os.path.join(REPO, "tests", "gpu_src", "decide_probe.hip")
    csrc = os.path.join(REPO, "pyphysim_amd", "csrc")
    subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
                    "-ffp-contract=fast", "-shared", "-I", csrc, "-I", os.path.join(REPO, "include"), src, "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    P = ctypes.c_void_p
    lib.probe_walk_decide.argtypes = [P, ctypes.c_int, P, P, ctypes.c_int, P, P]
    lib.probe_walk_decide.restype = ctypes.c_int
    return lib


def _argmin(table, pts):
    d = (pts.real[:, None] - table.real[None, :]) ** 2 + (pts.imag[:, None] - table.imag[None, :]) ** 2
    return np.argmin(d, axis=1).astype(np.int32)


def _points(table, seed, with_ties):
    """Probe points of a constellation: for every pair of points that share a decision boundary (each point's eight nearest
    neighbours), points along the perpendicular bisector (inside the segment and up to three spacings along the boundary) moved off
    it by +-2^-k of the spacing, k = 20 ... 42 (the certificates' margin is 2^-30; at 2^-42 the two squared distances still differ by
    thirty times their rounding, so numpy's plain and the kernel's fused evaluation order them alike); EXACT ties on the axes a
    constellation is exactly mirror-symmetric about (both distances are then the same floating-point expression: first index
    wins, in numpy and in the kernel); the constellation points themselves; rings far outside; random points."""
    rs = np.random.RandomState(seed)
    M = table.size
    dmin = np.min(np.abs(table[:, None] - table[None, :]) + 1e9 * np.eye(M))
    offs = [2.0 ** -k for k in (20, 29, 30, 31, 33, 38, 42)]
    pts = [table.copy()]
    for i in range(M):
        order = np.argsort(np.abs(table - table[i]))[1:9]
        for j in order:
            if j < i:
                continue
            mid, u = 0.5 * (table[i] + table[j]), (table[j] - table[i]) / abs(table[j] - table[i])
            for along in (0.0, 0.25, -0.4, 1.0, -3.0):
                base = mid + 1j * u * along * dmin
                for o in offs:
                    pts += [np.array([base + u * o * dmin, base - u * o * dmin])]
    ext = np.max(np.abs(table))
    if with_ties:
        t = rs.uniform(-1.2, 1.2, 600) * ext
        as_set = lambda v: set(zip(v.real.tolist(), v.imag.tolist()))
        if as_set(-np.conj(table)) == as_set(table):
            pts += [1j * t, np.zeros(1, dtype=complex)]                   # re = 0: a tie between c and -conj(c)
        if as_set(np.conj(table)) == as_set(table):
            pts += [t + 0j]                                               # im = 0: a tie between c and conj(c)
    ang = rs.uniform(0, 2 * np.pi, 400)
    for r in (1.5, 4.0, 50.0, 300.0, 1100.0):
        pts += [r * ext * np.exp(1j * ang)]
    pts += [(rs.uniform(-1.3, 1.3, 4000) + 1j * rs.uniform(-1.3, 1.3, 4000)) * ext]
    p = np.concatenate(pts).astype(np.complex128)
    return p[: 4 * (p.size // 4)]


def _run(probe, engine, method, pts, tx):
    n_groups = pts.size // 4
    pv = np.ascontiguousarray(pts.view(np.float64))
    tv = np.ascontiguousarray(tx.astype(np.int32))
    se = np.zeros(n_groups, dtype=np.uint32)
    be = np.zeros(n_groups, dtype=np.uint32)
    dec = probe.probe_walk_decide(engine.ctx, method, pv.ctypes.data, tv.ctypes.data, n_groups, se.ctypes.data, be.ctypes.data)
    assert dec >= 0, dec
    return dec, se, be


FORMS = [("qam", 16, _lib.CONST_QAM, _lib.DEMOD_MINDIST, "qam_cert"), ("qam", 64, _lib.CONST_QAM, _lib.DEMOD_MINDIST, "qam_cert"),
         ("qam", 256, _lib.CONST_QAM, _lib.DEMOD_MINDIST, "qam_cert"), ("qam", 4, _lib.CONST_QAM, _lib.DEMOD_MINDIST, "qam_cert"),
         ("qpsk", 4, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST, "quad_cert"), ("psk", 4, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST, "axis4_cert"),
         ("psk", 8, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST, "generic"), ("psk", 16, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST, "generic"),
         ("bpsk", 2, _lib.CONST_GENERIC, _lib.DEMOD_MINDIST, "generic"),
         ("qam", 16, _lib.CONST_QAM, _lib.DEMOD_QAM_SLICER, "slicer"), ("qam", 64, _lib.CONST_QAM, _lib.DEMOD_QAM_SLICER, "slicer"),
         ("qam", 256, _lib.CONST_QAM, _lib.DEMOD_QAM_SLICER, "slicer")]


@pytest.mark.parametrize("form", FORMS, ids=lambda f: "%s%d-%s" % (f[0], f[1], f[4]))
def test_every_decision_form_on_chosen_points(probe, engine, form):
    mod, M, kind, method, want_form = form
    table = np.asarray(chains.constellation(mod, M), dtype=np.complex128)
    engine.set_constellation(table, kind)
    # the slicer rounds half up where argmin takes the first index: its probe stays off the boundaries
    pts = _points(table, 11 + M, with_ties=want_form != "slicer")
    want = _argmin(table, pts)
    bits = np.array([bin(v).count("1") for v in range(256)], dtype=np.int64)
    dec, se, be = _run(probe, engine, method, pts, want)
    assert WDEC[dec] == want_form
    bad = np.flatnonzero(se)
    assert bad.size == 0, (bad[:5], pts[4 * bad[0]: 4 * bad[0] + 4], want[4 * bad[0]: 4 * bad[0] + 4])
    assert not be.any()
    # ... and the counts against labels that are NOT the decisions: every symbol / bit of the difference is counted
    rs = np.random.RandomState(5)
    tx = rs.randint(0, M, size=pts.size).astype(np.int32)
    _, se, be = _run(probe, engine, method, pts, tx)
    x = (tx ^ want).reshape(-1, 4)
    assert np.array_equal(se, (x != 0).sum(axis=1)) and np.array_equal(be, bits[x].sum(axis=1))


def test_the_probe_reaches_the_uncertified_branch(probe, engine):
    """With the certificates switched off (option demod_nocert) the same points take the table search everywhere: same counts --
    and with them on, a constellation whose decisions were wrong ONLY in the fallback would have failed above: the probe's points
    at 2^-31 ... 2^-42 of the spacing, on the axes and beyond the range are exactly the ones the certificates decline (modem.hpp: margin 2^-30)."""
    table = np.asarray(chains.constellation("qam", 64), dtype=np.complex128)
    engine.set_constellation(table, _lib.CONST_QAM)
    pts = _points(table, 3, with_ties=True)
    want = _argmin(table, pts)
    with engine.options(demod_nocert=1):
        dec, se, be = _run(probe, engine, _lib.DEMOD_MINDIST, pts, want)
    assert WDEC[dec] == "generic" and not se.any() and not be.any()
