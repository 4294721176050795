#!/usr/bin/env python3
"""Realizations/s of the planar MIMO-OFDM kernel family (csrc/pipeline_mimo_planar.hip) per geometry, next to the generic
radix-4 kernel k_run_mimo_ofdm<T, N, NA> it replaces (context option f64_generic): 64-QAM, cp 16, full band, SNR 25 dB,
min-distance demodulation (certificate) and slicer.  --dtype f64 (default) | f32; at (1024, 4x4) in f32 the matrix-core
kernel is timed too.  One JSON object on stdout (profiles/r04/f64_family_rates.json, f32_family_rates.json)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

DT = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "f64"
eng = Engine(0, DT)
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
nv = 10 ** -2.5
out = {}
for fft, nt, nr in ((256, 2, 2), (256, 4, 4), (512, 2, 2), (512, 4, 4), (1024, 2, 2), (1024, 4, 4), (2048, 2, 2), (2048, 4, 4),
                    (256, 2, 4), (1024, 2, 4)):
    n = max(16384, int(262144 * 4096 / (fft * nr)) // 8192 * 8192)
    n = min(n, 262144)
    row = {"realizations_per_launch": n}
    legs = [("fast_mindist", 0, _lib.DEMOD_MINDIST, 1), ("fast_slicer", 0, _lib.DEMOD_QAM_SLICER, 1),
            ("generic_mindist", 1, _lib.DEMOD_MINDIST, 1)]
    if DT == "f32" and (fft, nt, nr) == (1024, 4, 4):
        legs += [("mfma_mindist", 0, _lib.DEMOD_MINDIST, 0), ("mfma_slicer", 0, _lib.DEMOD_QAM_SLICER, 0)]
    if (DT == "f64" and ((nt == 4 and nr == 4) or (fft, nt, nr) == (256, 2, 2))) or (DT == "f32" and (fft, nt, nr) in ((256, 4, 4), (256, 2, 2))):
        # round 6: the default at the 4 x 4 shapes is the full-wave (256) / part-wave (512, 1024, 2048) kernel; the planar form of rounds 3-5 next to it
        legs += [("planar_mindist", 0, _lib.DEMOD_MINDIST, 261), ("planar_slicer", 0, _lib.DEMOD_QAM_SLICER, 261)]
    for name, generic, method, planar in legs:
        if generic and (nt != nr or (fft, nr) == (2048, 4)):      # the generic kernel has no such shape (2048 x 4: 181 KiB of LDS)
            continue
        cnt = eng.new_counters()
        with eng.options(f64_generic=generic, f32_mfma=0 if planar else 1, f64_threads=261 if planar == 261 else 0):
            run = lambda first: eng.run_mimo_ofdm(nt, nr, fft, 16, fft, 1, nv, 1, first, n, method=method, dtype=DT,
                                                  counters=cnt)
            _t0, _k = __import__('time').time(), 0        # untimed launches until the device has been busy for 60 ms (timed cold, short
            while _k < 64 and (_k < 1 or __import__('time').time() - _t0 < 0.06):   # launches read 5 - 10 % low: DESIGN 6, round 6)
                run((1 << 30) + _k * n)
                eng.sync()
                _k += 1
            eng.timer_start()
            for s in range(3):
                run(s * n)
            ms = eng.timer_stop_ms() / 3
        c = eng.read_counters(cnt)
        row[name] = dict(realizations_per_s=n / (ms * 1e-3), ms_per_launch=ms,
                         ser=c["sym_errors"] / float(c["n_realizations"] * fft * nt))
    if "generic_mindist" in row:
        row["fast_over_generic"] = row["fast_mindist"]["realizations_per_s"] / row["generic_mindist"]["realizations_per_s"]
    out["%dx%dx%d" % (fft, nt, nr)] = row
# per-subcarrier rate relative to the (1024, same geometry) row -- VERDICT r05 item 3's measure
for key, row in out.items():
    fft, nt, nr = (int(v) for v in key.split("x"))
    ref = out.get("1024x%dx%d" % (nt, nr))
    if ref:
        for leg in ("fast_mindist", "fast_slicer"):
            row[leg]["per_subcarrier_vs_1024"] = row[leg]["realizations_per_s"] * fft / (ref[leg]["realizations_per_s"] * 1024)
print(json.dumps(out, indent=1))
