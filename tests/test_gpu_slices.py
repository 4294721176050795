"""GPU: the solve + walk pipelines (IA, block diagonalisation, flat MIMO) work through their record buffer in slices of
2^20 realizations; a call that crosses the slice boundary gives the per-realization counts of two separate calls."""
import numpy as np
import pytest

from oracle import chains
from pyphysim_amd import _lib

pytestmark = pytest.mark.gpu
N = (1 << 20) + 77


def _same(whole, a, b):
    assert np.array_equal(whole[1], np.concatenate([a[1], b[1]])) and np.array_equal(whole[2], np.concatenate([a[2], b[2]]))
    for key in ("sym_errors", "bit_errors", "sym_errors_sq", "n_realizations", "n_skipped"):
        assert whole[0][key] == a[0][key] + b[0][key]


def test_ia_pipeline_across_the_slice_boundary(engine):
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    run = lambda first, count: engine.run_ia(4, 0.05, 99, first, count, method=_lib.DEMOD_QAM_SLICER, dtype="f32",
                                             per_realization=True)
    cut = (1 << 20) - 5
    _same(run(1000, N), run(1000, cut), run(1000 + cut, N - cut))


def test_bd_pipeline_across_the_slice_boundary(engine):
    engine.set_constellation(chains.constellation("psk", 4), _lib.CONST_GENERIC)
    run = lambda first, count: engine.run_bd(3, 2, 2, 1.0, 0.05, 7, first, count, method=_lib.DEMOD_MINDIST, dtype="f32",
                                             per_realization=True)
    cut = (1 << 20) + 3
    _same(run(0, N), run(0, cut), run(cut, N - cut))


def test_flat_mimo_pipeline_across_the_slice_boundary(engine):
    engine.set_constellation(chains.constellation("qam", 16), _lib.CONST_QAM)
    run = lambda first, count: engine.run_mimo_flat("alamouti", 2, 2, 2, 0.1, 5, first, count,
                                                    method=_lib.DEMOD_QAM_SLICER, dtype="f32", per_realization=True)
    cut = 1 << 20
    _same(run(50, N), run(50, cut), run(50 + cut, N - cut))


def test_mimo_ofdm_pipeline_across_its_slice_boundary(engine):
    """Config 4's matrix-core kernel reads its filter records in slices of 2^18 realizations."""
    engine.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    run = lambda first, count: engine.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, 0.003, 17, first, count, mmse=True,
                                                    method=_lib.DEMOD_QAM_SLICER, dtype="f32", per_realization=True)
    n, cut = (1 << 18) + 301, (1 << 18) - 2
    _same(run(9, n), run(9, cut), run(9 + cut, n - cut))
