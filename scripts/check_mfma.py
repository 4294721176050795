#!/usr/bin/env python3
"""GPU: the matrix-core config-4 kernel (csrc/pipeline_mimo_mfma.hip) against the VALU kernel and the oracle.
The context option no_mfma (mcle_ctx_set_option) selects the VALU kernel, so both run in one process on the same draws."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import chains, modem as omodem  # noqa: E402
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402

SEED = 20260927
eng = Engine(0, "f32")
out = {}


def run(first, count, mfma, method=_lib.DEMOD_QAM_SLICER, used=1024, nsym=1, cp=16, snr=25.0, mmse=True, per=True):
    eng.set_option("no_mfma", 0 if mfma else 1)
    eng.set_option("f32_mfma", 1 if mfma else 0)
    nv = 1.0 / (10.0 ** (snr / 10.0))
    return eng.run_mimo_ofdm(4, 4, 1024, cp, used, nsym, nv, SEED, first, count, mmse=mmse, method=method, dtype="f32",
                             per_realization=per)


eng.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
for name, kw in (("c4", {}), ("used600_2sym", dict(used=600, nsym=2)), ("cp7_zf", dict(cp=7, mmse=False, snr=30.0)),
                 ("mindist", dict(method=_lib.DEMOD_MINDIST))):
    n = 512
    r_new, se_new, be_new = run(1000, n, True, **kw)
    r_old, se_old, be_old = run(1000, n, False, **kw)
    nsymb = r_new["n_symbols"]
    out[name] = {"realizations": n, "n_symbols": nsymb,
                 "sum_se_new": int(se_new.astype(np.int64).sum()), "sum_se_old": int(se_old.astype(np.int64).sum()),
                 "max_abs_diff_per_realization": int(np.max(np.abs(se_new.astype(np.int64) - se_old.astype(np.int64)))),
                 "realizations_differing": int(np.count_nonzero(se_new != se_old)),
                 "bit_diff_total": int(np.abs(be_new.astype(np.int64) - be_old.astype(np.int64)).sum())}
    print(name, out[name], flush=True)
# oracle on a few realizations (f64 oracle vs f32 kernels)
kw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
want = np.array([chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **kw)["symbol_errors"] for r in range(1000, 1032)])
_, se_new, _ = run(1000, 32, True)
_, se_old, _ = run(1000, 32, False)
out["oracle32"] = {"oracle": want.tolist(), "mfma": se_new.tolist(), "valu": se_old.tolist()}
print("oracle ", want.tolist())
print("mfma   ", se_new.tolist())
print("valu   ", se_old.tolist())
# timing
VARIANTS = [v for v in os.environ.get("MFMA_VARIANTS", "").split(",") if v]
cases = [("valu_slicer", False, _lib.DEMOD_QAM_SLICER, None), ("mfma_slicer", True, _lib.DEMOD_QAM_SLICER, None),
         ("valu_mindist", False, _lib.DEMOD_MINDIST, None), ("mfma_mindist", True, _lib.DEMOD_MINDIST, None)]
cases += [("mfma_slicer_v" + v, True, _lib.DEMOD_QAM_SLICER, v) for v in VARIANTS]
cases += [("mfma_mindist_v" + v, True, _lib.DEMOD_MINDIST, v) for v in VARIANTS]
for name, mfma, method, variant in cases:
    cnt = eng.new_counters()
    eng.set_option("mfma_variant", int(variant or 0))
    eng.set_option("no_mfma", 0 if mfma else 1)
    eng.set_option("f32_mfma", 1 if mfma else 0)
    nv = 1.0 / (10.0 ** 2.5)
    for _ in range(3):
        eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 1 << 30, 65536, method=method, dtype="f32", counters=cnt)
    eng.sync()
    eng.timer_start()
    for s in range(10):
        eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, (1 << 31) + s * 65536, 65536, method=method, dtype="f32",
                          counters=cnt)
    ms = eng.timer_stop_ms() / 10
    out[name] = {"ms_per_launch": ms, "realizations_per_s": 65536 / ms * 1e3}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(REPO, "gpurun_out", "check_mfma.json"), "w"), indent=1)
