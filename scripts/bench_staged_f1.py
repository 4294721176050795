#!/usr/bin/env python
"""SURVEY.md section 8(f).1 at config-4 size, STAGED through HBM: 4x4 MMSE + 64-QAM + OFDM(1024, cp 16) over a
5-tap Jakes MIMO TDL channel, one receive filter per subcarrier.  Prints one JSON line with realizations/s and
the achieved HBM rate of the chain (algorithmic bytes of every stage / wall time).  Not the bench.py metric:
this chain is the operator-granular path, every intermediate makes a round trip through HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def staged_bytes(nt, nr, S, fft, cp, used, n_sym, dmax, cb=8):
    ns, n = used * n_sym, n_sym * (fft + cp)
    taps = cb * S * nr * nt * n
    b = 0
    b += 4 * nt * ns                                   # symbols written
    b += 4 * nt * ns + cb * nt * ns                    # modulate
    b += 2 * cb * nt * ns                              # Blast encode
    b += cb * nt * ns + cb * nt * n                    # OFDM modulate
    b += taps                                          # Jakes taps written
    b += taps + cb * nt * n + cb * nr * (n + dmax)     # TDL apply
    b += 2 * cb * nr * (n + dmax)                      # AWGN
    b += (2 * cb * nr * n) if dmax else 0              # drop the channel tail
    b += cb * nr * n + cb * nr * ns                    # OFDM demodulate
    b += taps + cb * ns * nr * nt                      # mean frequency response
    b += 2 * cb * ns * nr * nt                         # receive filters
    b += cb * ns * nr * nt + cb * nr * ns + cb * nt * ns   # per-subcarrier decode
    b += cb * nt * ns + 4 * nt * ns                    # demodulate + count
    return b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--fused", action="store_true", help="the fused kernel instead of the staged chain")
    ap.add_argument("--demod", default="slicer")
    args = ap.parse_args()
    from pyphysim_amd.simulators import MimoOfdmTdlSimulator
    Ts = 1.0 / (15e3 * 1024)
    sim = MimoOfdmTdlSimulator(SNR=[25.0], modulator="qam", M=64, Nt=4, Nr=4, fft_size=1024, cp_size=16,
                               num_ofdm_symbols=1, Fd=10.0, Ts=Ts, L=8, tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0),
                               tap_delays=np.arange(5) * Ts, dtype=args.dtype, batch_size=args.batch, seed=1,
                               fused=True if args.fused else False, demod=args.demod)
    p = next(iter(sim.params.get_unpacked_params_list()))
    sim._launch(p, 1 << 30, args.batch, False)
    sim.engine.sync()
    kernel_ms = None
    if args.fused:                      # kernel-only time (HIP events on the engine's stream), counters stay on device
        cnt = sim.engine.new_counters()
        run = lambda first: sim.engine.run_mimo_ofdm_tdl(
            4, 4, 1024, 16, 1024, 1, sim._noise_var(p), sim._tap_power, sim._tap_delay, sim._seed_for(p), first,
            args.batch, Fd=10.0, Ts=Ts, L=8, method=sim.demod_method, dtype=args.dtype, counters=cnt)
        run(1 << 31)
        sim.engine.sync()
        sim.engine.timer_start()
        for s in range(args.steps):
            run((1 << 32) + s * args.batch)
        kernel_ms = sim.engine.timer_stop_ms() / args.steps
    t0 = time.perf_counter()
    se = 0
    for s in range(args.steps):
        se += sim._launch(p, s * args.batch, args.batch, False)["sym_errors"]
    sim.engine.sync()
    dt = time.perf_counter() - t0
    n = args.steps * args.batch
    bytes_per = staged_bytes(4, 4, 5, 1024, 16, 1024, 1, 4, 8 if args.dtype == "f32" else 16)
    print(json.dumps({"workload": "4x4 MMSE + 64-QAM + OFDM(1024,16) over 5-tap Jakes MIMO TDL (%s, 8f.1)"
                                  % ("fused" if args.fused else "staged"),
                      "realizations_per_s": n / dt, "batch": args.batch, "dtype": args.dtype,
                      "ser": se / float(n * 4096), "staged_bytes_per_realization": bytes_per,
                      "achieved_GBps": bytes_per * n / dt / 1e9,
                      "kernel_ms_per_launch": kernel_ms,
                      "kernel_realizations_per_s": (args.batch / (kernel_ms * 1e-3)) if kernel_ms else None}))


if __name__ == "__main__":
    main()
