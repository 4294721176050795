"""Mirror of pyphysim.comm.waterfilling (reference comm/waterfilling.py:15-92)."""
import numpy as np

from ..engine import get_engine

__all__ = ["doWF"]


def doWF(vtChannels, dPt, noiseVar=1.0, Es=1.0, engine=None):
    """Water-filling over parallel AWGN channels with POWER gains ``vtChannels``: returns
    ``(vtOptP, mu)`` -- the optimum powers (input order) and the water level.  ``Es`` scales the gains as
    in the reference (level = noiseVar / (Es * gain)); the water level is reported on the unscaled gains
    like the reference's last line (waterfilling.py:90)."""
    eng = engine if engine is not None else get_engine()
    g = np.asarray(vtChannels, dtype=float)
    P, mu = eng.waterfilling(g * float(Es), dPt, noiseVar)
    if Es != 1.0:
        mu = float(P[np.argmax(g)] + float(noiseVar) / np.max(g))
    return P, mu
