"""round 5: which form of the library's streaming kernel reaches the highest HBM rate on this box"""
import sys; sys.path.insert(0, '.')
from pyphysim_amd.engine import Engine
eng = Engine(0, "f32")
for kind in ("copy", "read", "triad", "write"):
    for nt in (False, True):
        for u8 in (False, True):
            row = []
            for bpc in (4, 8, 16, 32, 64):
                row.append("%5.0f" % eng.hbm_stream_rate(kind, 1 << 30, 10, bpc, nt, u8))
            print("%-6s nt=%d u8=%d  bpc 4/8/16/32/64: %s" % (kind, nt, u8, " ".join(row)))
for nb in (1 << 28, 1 << 31):
    print("copy nbytes", nb, "%5.0f" % eng.hbm_stream_rate("copy", nb, 10, 16, True, False))
