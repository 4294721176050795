#!/usr/bin/env python3
"""HBM throughput of the per-operator (staged) kernels on device-resident arrays: algorithmic bytes
(inputs read once + outputs written once) / HIP-event time.  Run on the GPU box:
    python scripts/bench_operators.py > gpurun_out/operators.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyphysim_amd import _lib  # noqa: E402
from pyphysim_amd.engine import Engine  # noqa: E402
from pyphysim_amd.modulators import constellation  # noqa: E402

DT = "f64" if "--dtype=f64" in sys.argv or os.environ.get("BENCH_OPS_DTYPE") == "f64" else "f32"
CB = 16 if DT == "f64" else 8     # bytes per complex sample
eng = Engine(0, DT)
eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
N = (1 << 25) if DT == "f64" else (1 << 26)   # 512 MiB of samples either way, past the 256 MiB Infinity Cache
REP = 5
rows = []


def timed(name, fn, nbytes):
    fn()
    eng.sync()
    eng.timer_start()
    for _ in range(REP):
        fn()
    ms = eng.timer_stop_ms() / REP
    rows.append({"kernel": name, "ms": ms, "algorithmic_GB": nbytes / 1e9, "GBps": nbytes / ms / 1e6,
                 "frac_of_8TBps": nbytes / ms / 1e6 / 8000.0})


ONLY = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]


def jakes_rows(cases):
    """The Jakes generator alone: S streams of N / S samples, L rays, written once (no input but 2 L S phases)."""
    rs = np.random.RandomState(5)
    for L, S in cases:
        n = N // S
        phi, psi = rs.uniform(0, 2 * np.pi, (L, S)), rs.uniform(0, 2 * np.pi, (L, S))
        timed("jakes_generate L=%d, %d streams (-> %d B)" % (L, S, CB),
              lambda: eng.jakes_generate(phi, psi, 100.0, 1e-3, 1e-3, n, device=True), N * CB)


if ONLY == ["jakes"]:
    jakes_rows(((8, 16), (16, 16), (12, 16)))
    print(json.dumps({"device": eng.device_name, "dtype": DT, "bytes_per_sample": CB, "n_symbols": N, "rows": rows}, indent=1))
    sys.exit(0)
idx = eng.rand_symbols(N, 64, 1, 2, device=True)
tx = eng.modulate(idx)
noise = eng.randn_c(N, 1, 2, device=True)
rx = eng.awgn_add(tx, noise, 0.01)
cnt = eng.new_counters()          # device-resident: the counting kernels are timed without a read-back
timed("modulate (4 B idx -> 8 B sample)", lambda: eng.modulate(idx), N * (4 + CB))
timed("awgn_add (8+8 -> 8)", lambda: eng.awgn_add(tx, noise, 0.01), N * 3 * CB)
timed("demodulate slicer (8 -> 4)", lambda: eng.demodulate(rx, method=_lib.DEMOD_QAM_SLICER), N * (4 + CB))
timed("demodulate mindist M=64 (8 -> 4)", lambda: eng.demodulate(rx), N * (4 + CB))
timed("demod_count slicer (8 + 4 -> counters)", lambda: eng.demod_count(rx, idx, n_real=1024, method=_lib.DEMOD_QAM_SLICER, counters=cnt), N * (4 + CB))
dec = eng.demodulate(rx, method=_lib.DEMOD_QAM_SLICER)
timed("count_errors (4 + 4 -> counters)", lambda: eng.count_errors(idx, dec, 6, n_real=1024, counters=cnt), N * 8)
timed("cdiv (8+8 -> 8)", lambda: eng.cdiv(rx, tx), N * 3 * CB)
timed("randn_c Philox (-> 8)", lambda: eng.randn_c(N, 1, 2, device=True), N * CB)
jakes_rows(((8, 16), (16, 16)))
nsym = N // 1024
timed("ofdm_modulate 1024+16 (8 -> 8.1)", lambda: eng.ofdm_modulate(tx, 1024, 16, 1024), N * CB + nsym * 1040 * CB)
t = eng.ofdm_modulate(tx, 1024, 16, 1024)
timed("ofdm_demodulate 1024+16", lambda: eng.ofdm_demodulate(t, 1024, 16, 1024), N * CB + nsym * 1040 * CB)
b = 4096
ns = N // (4 * b)
X = tx.reshape(b, 4, ns)
H = eng.randn_c(b * 16, 3, 4, device=True).reshape(b, 4, 4)
timed("mimo_channel 4x4 (H X)", lambda: eng.mimo_channel(H, X), b * 4 * ns * 2 * CB)
G, _ = eng.blast_filter(H, 0.01)
timed("blast_decode 4x4 (G Y)", lambda: eng.blast_decode(G, X), b * 4 * ns * 2 * CB)
timed("blast_encode (transpose)", lambda: eng.blast_encode(tx, 4, batch=b), N * 2 * CB)
# ---- channel-side operators (a9 / a10) and the remaining MIMO schemes (a13) ------------------------------------------------
n5 = N // 8
x5 = eng.randn_c(n5, 13, 14, device=True)
delays5 = np.arange(5, dtype=np.int32)
taps5 = eng.randn_c(5 * n5, 5, 6, device=True).reshape(5, n5)
timed("tdl_apply 5 taps (x + 5 taps -> y)", lambda: eng.tdl_apply(x5, taps5, delays5), n5 * 7 * CB)
nsym5 = n5 // 1040
d5 = eng.randn_c(nsym5 * 1024, 15, 16, device=True)
t5 = eng.randn_c(5 * nsym5 * 1040, 17, 18, device=True).reshape(5, nsym5 * 1040)
timed("onetap_equalize 1024+16, 5 taps (data + taps -> data)", lambda: eng.onetap_equalize(d5, t5, delays5, 1024, 16, 1024),
      nsym5 * (2 * 1024 + 5 * 1040) * CB)
timed("tdl_mean_freq_response 1024+16, 5 taps (taps -> H)", lambda: eng.tdl_mean_freq_response(t5, delays5, nsym5, 1024, 16, 1024),
      nsym5 * (1024 + 5 * 1040) * CB)
ba, na = 4096, N // (4 * 4096)
xa = eng.randn_c(ba * 2 * na, 19, 20, device=True)
Ha = eng.randn_c(ba * 4, 7, 8, device=True).reshape(ba, 2, 2)
timed("alamouti_encode (x -> 2 antennas)", lambda: eng.alamouti_encode(xa, batch=ba), ba * 2 * na * 3 * CB)
Ya = eng.alamouti_encode(xa, batch=ba)
timed("alamouti_decode 2x2 (Y -> x)", lambda: eng.alamouti_decode(Ha, Ya), ba * 2 * na * 3 * CB)
hm = eng.randn_c(ba * 4, 9, 10, device=True).reshape(ba, 4)
xm = eng.randn_c(ba * na, 21, 22, device=True)
timed("mrt_encode 4 antennas (x -> 4 x)", lambda: eng.mrt_encode(hm, xm), ba * na * 5 * CB)
Hb = eng.randn_c((1 << 20) * 16, 11, 12, device=True).reshape(1 << 20, 4, 4)
timed("blast_filter 4x4 MMSE (H -> G), 2^20 matrices", lambda: eng.blast_filter(Hb, 0.01, read_skipped=False),
      (1 << 20) * 32 * CB)
print(json.dumps({"device": eng.device_name, "dtype": DT, "bytes_per_sample": CB, "n_symbols": N, "rows": rows}, indent=1))
