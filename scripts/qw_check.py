"""Quarter-wave config-4 kernel (csrc/pipeline_mimo_qw.hip, option f64_threads = 260) against the planar kernel and the oracle:
per-realization counts, then the rate of both.  Runs on the GPU box: python scripts/qw_check.py [n_check] [batch]"""
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


def main():
    n_check = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    eng = Engine(0, "f64")
    eng.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
    nv = 1.0 / float(omodem.dB2Linear(25.0))
    first = 1000
    ref = eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, first, n_check, dtype="f64", per_realization=True)
    with eng.options(f64_threads=260):
        qw = eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, first, n_check, dtype="f64", per_realization=True)
    print("planar se[:8]", ref[1][:8].tolist())
    print("qw     se[:8]", qw[1][:8].tolist())
    print("equal counts:", bool(np.array_equal(ref[1], qw[1]) and np.array_equal(ref[2], qw[2])), flush=True)
    kw = dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1, snr_db=25.0, mmse=True)
    want = [chains.chain_mimo_ofdm(chains.PhiloxRng(SEED, r), **kw) for r in range(first, first + min(n_check, 16))]
    wse = np.array([w["symbol_errors"] for w in want])
    print("oracle se[:8]", wse[:8].tolist(), "qw == oracle:", bool(np.array_equal(wse, qw[1][:len(wse)])), flush=True)
    for name, opt in (("planar", 0), ("qw", 260), ("qw_wps2", 262)):
        with eng.options(f64_threads=opt):
            cnt = eng.new_counters()
            eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 1 << 30, batch, dtype="f64", counters=cnt)
            eng.sync()
            eng.timer_start()
            for s in range(3):
                eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, (1 << 31) + s * batch, batch, dtype="f64", counters=cnt)
            ms = eng.timer_stop_ms() / 3
            c = eng.read_counters(cnt)
            print("%-8s %.3f ms per %d  -> %.4e realizations/s   ser %.6f" % (name, ms, batch, batch / ms * 1e3,
                  c["sym_errors"] / (c["n_realizations"] * 4096.0)), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
