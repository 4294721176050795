"""GPU: the planar MIMO-OFDM kernel family in complex64 (csrc/pipeline_mimo_planar.hip, k_run_mimo_ofdm_planar<float, ...>: the
complex128 kernels of test_gpu_f64_kernel.py on planes of floats -- radix-4 stages at fft_size 256 .. 2048 for every
1 <= Nt <= Nr <= 4, radix-16 register passes with the fused channel stage at (1024, Nr = 4)).

complex64 statements of the link round differently from the complex128 oracle, so per-realization counts may differ by a
boundary tie now and then, never systematically: |dSER| <= 1e-4 (the north star's tolerance), ties <= 3 per realization --
the criteria of the matrix-core kernel's tests (test_gpu_mfma.py), which this family complements (every geometry but
(1024, 4x4) had only the round-1 generic kernel in complex64) and, since round 4, replaces at (1024, 4x4) by default
(option f32_mfma = 1 selects the matrix-core kernel)."""
import numpy as np
import pytest

from oracle import chains, modem as omodem
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
SEED = 515151

SHAPES = [(256, 2, 2), (256, 4, 4), (512, 2, 2), (512, 4, 4), (1024, 2, 2), (1024, 4, 4), (2048, 2, 2), (2048, 4, 4),
          (1024, 2, 4), (256, 2, 4), (512, 1, 2), (1024, 3, 4), (1024, 1, 4), (2048, 1, 4), (256, 2, 3), (1024, 3, 3), (512, 1, 3),
          (2048, 3, 4), (2048, 2, 4)]
SHAPE_CASES = [dict(mod="qam", M=64, snr_db=25.0),
               dict(mod="qam", M=16, snr_db=17.0, used_frac=0.6, n_ofdm_sym=2, cp_size=7, mmse=False),   # partial band, odd CP, ZF
               dict(mod="psk", M=8, snr_db=13.0, n_ofdm_sym=2, cp_size=33)]                               # candidate grid only


def _okw(kw, fft, nt, nr):
    used = fft if "used_frac" not in kw else 2 * int(kw["used_frac"] * fft / 2)
    return dict(mod=kw["mod"], M=kw["M"], nt=nt, nr=nr, fft_size=fft, cp_size=kw.get("cp_size", 16), num_used=used,
                n_ofdm_sym=kw.get("n_ofdm_sym", 1), snr_db=kw["snr_db"], mmse=kw.get("mmse", True))


def _run(engine, okw, first, count, method, threads=0, planar=1, generic=0, dtype="f32"):
    nv = 1.0 / omodem.dB2Linear(okw["snr_db"]) if okw["snr_db"] is not None else 0.0
    with engine.options(f32_mfma=0 if planar else 1, f64_threads=threads, f64_generic=generic):
        return engine.run_mimo_ofdm(okw["nt"], okw["nr"], okw["fft_size"], okw["cp_size"], okw["num_used"], okw["n_ofdm_sym"],
                                    nv, SEED, first, count, mmse=okw["mmse"], method=method, dtype=dtype, per_realization=True)


def _oracle(okw, first, count):
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **okw) for r in range(first, first + count)]
    return (np.array([w["symbol_errors"] for w in want], dtype=np.int64), np.array([w["bit_errors"] for w in want], dtype=np.int64),
            want[0]["num_symbols"], want[0]["num_bits"])


def _close(se, be, want_se, want_be, nsym, nbits, tag):
    ok = se != 0xFFFFFFFF                                                    # (a skipped realization: singular filter in f32)
    assert np.count_nonzero(~ok) <= 1, tag
    n = int(ok.sum())
    assert abs(int(se[ok].astype(np.int64).sum()) - int(want_se[ok].sum())) <= 1e-4 * n * nsym + 2, tag
    assert abs(int(be[ok].astype(np.int64).sum()) - int(want_be[ok].sum())) <= 1e-4 * n * nbits + 2, tag
    assert np.max(np.abs(se[ok].astype(np.int64) - want_se[ok])) <= 3, (tag, se, want_se)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%dx%d" % s)
@pytest.mark.parametrize("case", range(len(SHAPE_CASES)))
def test_f32_family_against_the_oracle(engine, shape, case):
    fft, nt, nr = shape
    okw = _okw(SHAPE_CASES[case], fft, nt, nr)
    kind = _lib.CONST_QAM if okw["mod"] == "qam" else _lib.CONST_GENERIC
    engine.set_constellation(chains.constellation(okw["mod"], okw["M"]), kind)
    first, count = (1 << 34) + 11, 10
    want_se, want_be, nsym, nbits = _oracle(okw, first, count)
    methods = [_lib.DEMOD_MINDIST] + ([_lib.DEMOD_QAM_SLICER] if okw["mod"] == "qam" else [])
    forms = (0, 257, 259, 512, 256) if shape == (1024, 4, 4) else (0,)      # radix-16 unfused (4 / 3 wavefronts per SIMD) / fused, radix-4
    for method in methods:
        for threads in forms:
            res, se, be = _run(engine, okw, first, count, method, threads=threads)
            assert res["n_symbols"] == nsym and res["n_bits"] == nbits and res["n_realizations"] + res["n_skipped"] == count
            _close(se, be, want_se, want_be, nsym, nbits, (shape, case, method, threads))
            ok = se != 0xFFFFFFFF
            assert res["sym_errors"] == int(se[ok].astype(np.int64).sum()) and res["bit_errors"] == int(be[ok].astype(np.int64).sum())
    if nt == nr and nr != 3:                                                # against the generic complex64 kernel of the same shape
        n = 300
        new, se, be = _run(engine, okw, 9, n, _lib.DEMOD_MINDIST)
        old, se_o, be_o = _run(engine, okw, 9, n, _lib.DEMOD_MINDIST, generic=1, planar=1)
        ok = (se != 0xFFFFFFFF) & (se_o != 0xFFFFFFFF)
        assert np.count_nonzero(~ok) <= 1
        assert np.max(np.abs(se[ok].astype(np.int64) - se_o[ok].astype(np.int64))) <= 3
        assert abs(int(se[ok].astype(np.int64).sum()) - int(se_o[ok].astype(np.int64).sum())) <= 1e-4 * n * nsym + 2


def test_f32_planar_headline_geometry_aggregate_ser(engine):
    """(1024, 4x4), 64-QAM, 25 dB on 2 048 realizations = 8.4e6 symbols: |dSER| <= 1e-5 against the oracle, no sign, both
    demodulators; and against the matrix-core kernel on the same draws."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    okw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=1024, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    first, n = 70000, 2048
    want_se, want_be, nsym, nbits = _oracle(okw, first, n)
    for method in (_lib.DEMOD_QAM_SLICER, _lib.DEMOD_MINDIST):
        res, se, be = _run(engine, okw, first, n, method)
        d = se.astype(np.int64) - want_se
        assert abs(int(d.sum())) <= 1e-5 * n * nsym, (method, int(d.sum()))
        assert abs(int((be.astype(np.int64) - want_be).sum())) <= 1e-5 * n * nbits
        assert np.max(np.abs(d)) <= 3 and np.count_nonzero(d) <= 0.05 * n
        assert res["n_realizations"] == n and res["n_skipped"] == 0
        _, se_m, _ = _run(engine, okw, first, n, method, planar=0)          # the matrix-core kernel
        assert np.max(np.abs(se.astype(np.int64) - se_m.astype(np.int64))) <= 3


def test_f32_planar_is_split_invariant_and_clean_without_noise(engine):
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    for fft, nt, nr in ((1024, 4, 4), (512, 2, 4), (2048, 2, 2)):
        okw = dict(mod="qam", M=64, nt=nt, nr=nr, fft_size=fft, cp_size=16, num_used=fft, n_ofdm_sym=1, snr_db=25.0, mmse=True)
        whole, se, be = _run(engine, okw, 1 << 32, 700, _lib.DEMOD_MINDIST)
        again, se2, be2 = _run(engine, okw, 1 << 32, 700, _lib.DEMOD_MINDIST)
        assert np.array_equal(se, se2) and np.array_equal(be, be2) and whole == again
        a = _run(engine, okw, 1 << 32, 123, _lib.DEMOD_MINDIST)
        b = _run(engine, okw, (1 << 32) + 123, 577, _lib.DEMOD_MINDIST)
        assert np.array_equal(np.concatenate([a[1], b[1]]), se) and np.array_equal(np.concatenate([a[2], b[2]]), be)
        for k in ("sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq", "n_realizations", "n_skipped"):
            assert whole[k] == a[0][k] + b[0][k], k
        clean = dict(okw, snr_db=None, mmse=False)
        res = _run(engine, clean, 5, 2048, _lib.DEMOD_QAM_SLICER)[0]
        assert res["n_realizations"] > 2000 and res["sym_errors"] <= 1e-6 * res["n_realizations"] * fft * nt + 8
