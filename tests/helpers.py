"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases(name):
    """Yield (kwargs, realization dicts) for every stored case of a chain fixture."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = []
    for ci in range(int(z["n_cases"])):
        kw = json.loads(str(z["case%d_kwargs" % ci]))
        for k, v in list(kw.items()):
            if isinstance(v, list):
                kw[k] = tuple(v)
        reals = []
        for r in range(int(z["n_real"])):
            pre = "case%d_r%d_" % (ci, r)
            reals.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        out.append((kw, reals))
    return out


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    return float(np.max(np.abs(a - b))) / scale if a.size else 0.0
