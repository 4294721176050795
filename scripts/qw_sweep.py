"""Quarter-wave kernel rate against the persistent grid's oversubscription (option grid_oversub) and the batch; GPU box."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import chains, modem as omodem  # noqa: E402
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402

eng = Engine(0, "f64")
eng.set_constellation(chains.constellation("qam", 64), _lib.CONST_QAM)
nv = 1.0 / float(omodem.dB2Linear(25.0))
for batch in (262144,):
    for g in (0, 1, 16):
        with eng.options(grid_oversub=g):
            cnt = eng.new_counters()
            eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, 7, 1 << 30, batch, dtype="f64", counters=cnt)
            eng.sync()
            eng.timer_start()
            for s in range(3):
                eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, 7, (1 << 31) + s * batch, batch, dtype="f64", counters=cnt)
            ms = eng.timer_stop_ms() / 3
            print("batch %8d oversub %2d: %.3f ms -> %.4e /s" % (batch, g, ms, batch / ms * 1e3), flush=True)
eng.close()
