#!/usr/bin/env python
"""BASELINE config 4 STAGED through HBM -- the operator-granular form the north star's "coalesced HBM for the symbol
streams ... at >= 40 % of the HBM roofline" clause is quoted on (every operator reads its input
arrays once and writes its output once, B_alg = 412 160 B per realization in complex64):

    rand_modulate_batch (gen + modulate) -> blast_encode -> ofdm_modulate x4 -> randn_c_batch (H) -> mimo_channel_philox
    (H T + noise) -> ofdm_demodulate x4 -> blast_filter -> blast_decode -> demod_count (demodulate + count)

device-resident, a batch of realizations per kernel, the draws of realization r exactly those of the fused kernel (same
mcle-philox-v1 ledger: DATA symbols, CHAN H row-major, NOISE (Nr, 1040) row-major), so the error counts equal
mcle_run_mimo_ofdm's on the same indices (tests/test_gpu_staged_c4.py).  Prints one JSON line: realizations/s, achieved
algorithmic GB/s = B_alg x rate, fraction of 8 TB/s and of the measured copy bandwidth.  `run()` is what bench.py calls for
its other_workloads.c4_staged leg."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SEED = 20260927
SNR_DB = 25.0
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md
HBM_COPY_GBPS = 6290.0      # same guide: measured float4 copy
# SURVEY.md section 8(d), config 4, complex64 samples (S = 8 B) and byte indices (I = 1 B)
B_ALG = {"H + MMSE filter": 384, "symbols + modulate": 36864, "Blast encode": 65536, "OFDM modulate x4": 66048,
         "H T + awgn": 66688, "OFDM demodulate x4": 66048, "Blast decode": 65536, "demodulate": 36864, "count": 8192}


def chain(eng, first, count, counters, dtype="f32", method=None, noise_var=None, seed=SEED, M=64):
    """One pass of the staged chain over realizations [first, first + count); adds into the device counter block."""
    from pyphysim_amd import _lib
    method = _lib.DEMOD_MINDIST if method is None else method
    nv = (1.0 / (10.0 ** (SNR_DB / 10.0))) if noise_var is None else noise_var
    idx, sym = eng.rand_modulate_batch(4096, seed, first, count, dtype=dtype)               # [count, 4096] int32, complex
    X = eng.blast_encode(sym, 4, batch=count, dtype=dtype)                                  # [count, 4, 1024]
    T = eng.ofdm_modulate(X, 1024, 16, 1024, batch=count * 4, dtype=dtype).reshape(count, 4, 1040)
    H = eng.randn_c_batch(16, seed, first, count, stream=_lib.STREAM_CHAN, dtype=dtype).reshape(count, 4, 4)
    R = eng.mimo_channel_philox(H, T, seed, first, nv, dtype=dtype)                         # [count, 4, 1040]
    Y = eng.ofdm_demodulate(R, 1024, 16, 1024, batch=count * 4, dtype=dtype).reshape(count, 4, 1024)
    G, _ = eng.blast_filter(H, nv, dtype=dtype, read_skipped=False)
    est = eng.blast_decode(G, Y, dtype=dtype)                                               # [count, 4096]
    return eng.demod_count(est, idx, n_real=count, method=method, dtype=dtype, counters=counters)


def bind(eng, M=64):
    from pyphysim_amd import _lib
    from pyphysim_amd.modulators import constellation
    eng.set_constellation(constellation("qam", M), _lib.CONST_QAM)


def run(eng, batch=8192, seconds=1.0, dtype="f32", demod="mindist"):
    """Time the chain for about `seconds` (HIP events around whole passes) -> the dict bench.py embeds."""
    from pyphysim_amd import _lib
    method = _lib.DEMOD_MINDIST if demod == "mindist" else _lib.DEMOD_QAM_SLICER
    bind(eng)
    cnt = eng.new_counters()
    for w in range(3):                                   # buffers into the engine's pool, clocks up
        chain(eng, (1 << 43) + w * batch, batch, cnt, dtype, method)
    eng.sync()
    t0 = time.perf_counter()
    chain(eng, 1 << 44, batch, cnt, dtype, method)
    eng.sync()
    one = max(time.perf_counter() - t0, 1e-4)
    steps = max(3, min(2000, int(seconds / one)))
    cnt.zero()
    eng.sync()
    t0 = time.perf_counter()
    eng.timer_start()
    for s in range(steps):
        chain(eng, s * batch, batch, cnt, dtype, method)
    ms = eng.timer_stop_ms()
    wall = time.perf_counter() - t0
    c = eng.read_counters(cnt)
    n = steps * batch
    rate = n / (ms * 1e-3)
    balg = sum(B_ALG.values()) * (2 if dtype == "f64" else 1)
    return {"workload": "config 4 staged through HBM, one kernel per reference operator (SURVEY 8(d) staged model)",
            "dtype": dtype, "demod": demod, "realizations_per_s": rate, "batch": batch, "passes": steps,
            "ms_per_pass": ms / steps, "wall_s": wall, "b_alg_bytes_per_realization": balg, "b_alg_breakdown": B_ALG,
            "achieved_GBps": balg * rate / 1e9, "frac": balg * rate / 1e9 / HBM_PEAK_GBPS,
            "frac_of_measured_copy_bw": balg * rate / 1e9 / HBM_COPY_GBPS, "peak_GBps": HBM_PEAK_GBPS,
            "north_star_clause": ">= 1e6 realizations/s at >= 40 %% of the HBM roofline: %s" % (
                "met" if rate >= 1e6 and balg * rate / 1e9 / HBM_PEAK_GBPS >= 0.40 else "not met"),
            "ser": c["sym_errors"] / float(max(1, c["n_realizations"]) * 4096), "n_realizations": c["n_realizations"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--demod", default="mindist", choices=["mindist", "slicer"])
    args = ap.parse_args()
    from pyphysim_amd.engine import Engine
    eng = Engine(0, args.dtype)
    print(json.dumps(run(eng, args.batch, args.seconds, args.dtype, args.demod)))
    eng.close()


if __name__ == "__main__":
    main()
