#!/usr/bin/env python3
"""bench.py -- Monte Carlo realizations/s of the fused HIP link pipeline on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json north_star target, configs[3] geometry on ONE GPU per rank):
  4x4 MIMO (Blast, MMSE) + 64-QAM + OFDM-1024 (cp 16, all bins used), flat H ~ randn_c(4,4) per
  realization, SNR 25 dB; a "step" = one batch of --batch realizations per GPU through
  mcle_run_mimo_ofdm (data, channel and noise drawn on-chip from (seed, realization index)).
  Realization index ranges are disjoint across ranks and steps ("weak" scaling); the only
  exchange is one all-reduce (RCCL) of the 8-word integer counter block at the end.
"""
import argparse
import json
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # before NumPy loads its BLAS:
    os.environ.setdefault(_v, "1")                                        # the CPU legs are per-core figures

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# algorithmic bytes per realization in the staged (operator-granular)
# model, complex64 samples / uint8 indices.
B_ALG = {"c2": 7_600_000, "c3": 211_360, "c4": 412_160, "c5": 62_000, "f1": 3_099_008,
         # f6 (6 streams x 500 symbols): gen+mod 27 000; precode, channel, filter 48 000 each; demod 27 000; count 6 000
         "f6": 204_000}
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
SEED = 20260927
SNR_DB = {"c1": 10.0, "c2": 20.0, "c3": 20.0, "c4": 25.0, "c5": 20.0, "f1": 25.0, "f6": 15.0}
F1_TS = 1.0 / (15e3 * 1024)
F1_TAPS_DB = (0.0, -3.0, -6.0, -9.0, -12.0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5", "f1", "f6"])
    ap.add_argument("--batch", type=int, default=0, help="realizations per GPU per step")
    ap.add_argument("--demod", default="slicer", choices=["slicer", "mindist"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--cpu-multicore-seconds", type=float, default=6.0,
                    help="budget of the all-cores CPU leg (0 disables it)")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def make_runner(eng, cfg, demod, dtype):
    """-> (run(first, count, counters), units per realization, description)"""
    from pyphysim_amd.modulators import constellation  # product-side tables (no oracle import here)
    from pyphysim_amd import _lib
    method = _lib.DEMOD_QAM_SLICER if demod == "slicer" else _lib.DEMOD_MINDIST
    nv = 1.0 / (10.0 ** (SNR_DB[cfg] / 10.0))
    if cfg == "c4":
        eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)

        def run(first, count, counters):
            eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, first, count, mmse=True, method=method,
                              dtype=dtype, counters=counters)
        return run, 4096, "4x4 Blast-MMSE + 64-QAM + OFDM(1024, cp 16), flat randn_c H, SNR 25 dB (config 4)"
    if cfg == "c2":
        eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)

        def run(first, count, counters):
            eng.run_flat_fading(100000, nv, SEED, first, count, Fd=100.0, Ts=1e-3, L=8, method=method, dtype=dtype,
                                counters=counters)
        return run, 100000, "64-QAM over flat Jakes fading (Fd 100 Hz, Ts 1 ms, L 8), 1e5 symbols, SNR 20 dB (config 2)"
    if cfg == "f1":
        eng.set_constellation(constellation("qam", 64), _lib.CONST_QAM)
        from pyphysim_amd.channels import discretize_profile
        p_lin, d_idx = discretize_profile(np.array(F1_TAPS_DB), np.arange(5) * F1_TS, F1_TS)

        def run(first, count, counters):
            eng.run_mimo_ofdm_tdl(4, 4, 1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, first, count, Fd=10.0, Ts=F1_TS,
                                  L=8, mmse=True, method=method, dtype=dtype, counters=counters)
        return run, 4096, ("4x4 MMSE per subcarrier + 64-QAM + OFDM(1024, cp 16) over a 5-tap Jakes MIMO TDL channel "
                           "(Fd 10 Hz), SNR 25 dB (SURVEY 8(f).1)")
    if cfg == "f6":
        eng.set_constellation(constellation("psk", 4), _lib.CONST_GENERIC)

        def run(first, count, counters):
            eng.run_bd(3, 2, 500, 1.0, nv, SEED, first, count, method=_lib.DEMOD_MINDIST, dtype=dtype,
                       counters=counters)
        return run, 3000, ("K=3 cells of 2x2 antennas, block-diagonalising precoder (water-filling) + zero forcing, "
                           "4-PSK, 500 symbols/stream, SNR 15 dB (apps/comp_BD/simulate_comp_simple.py; SURVEY 8(f).3)")
    if cfg == "c5":
        eng.set_constellation(constellation("qam", 16), _lib.CONST_QAM)

        def run(first, count, counters):
            eng.run_ia(200, nv, SEED, first, count, method=method, dtype=dtype, counters=counters)
        return run, 600, "K=3 2x2 closed-form interference alignment + 16-QAM, 200 symbols/stream, SNR 20 dB (config 5)"
    eng.set_constellation(constellation("qpsk", 4), _lib.CONST_GENERIC)
    from pyphysim_amd.channels import discretize_profile
    Ts = 1.0 / (15e3 * 1024)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)

    def run(first, count, counters):
        eng.run_ofdm_tdl(1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, first, count, Fd=10.0, Ts=Ts, L=8,
                         method=_lib.DEMOD_MINDIST, dtype=dtype, counters=counters)
    return run, 1024, "QPSK + OFDM(1024, cp 16) over 5-tap Jakes TDL (Fd 10 Hz), one-tap EQ, SNR 20 dB (config 3)"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _oracle_chain(cfg):
    """(oracle chain, kwargs) of a bench configuration -- used by the cpu_baseline legs only."""
    from oracle import chains
    return {
        "c4": (chains.chain_mimo_ofdm, dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None,
                                           n_ofdm_sym=1, snr_db=25.0, mmse=True)),
        "c2": (chains.chain_flat_jakes, dict(mod="qam", M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8)),
        "c3": (chains.chain_ofdm_tdl, dict(mod="qpsk", M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1,
                                           snr_db=20.0, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8)),
        "f1": (chains.chain_mimo_ofdm_tdl, dict(mod="qam", M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None,
                                                n_ofdm_sym=1, snr_db=25.0, Fd=10.0, Ts=F1_TS, L=8,
                                                tap_powers_dB=F1_TAPS_DB, tap_delays_samples=(0, 1, 2, 3, 4))),
        "c5": (chains.chain_ia, dict(mod="qam", M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0)),
        "f6": (chains.chain_bd, dict(mod="psk", M=4, K=3, nr=2, NSymbs=500, iPu=1.0,
                                     noise_var=1.0 / (10.0 ** (SNR_DB["f6"] / 10.0)), canonical=True)),
    }[cfg]


def cpu_baseline(cfg, budget_s, gpu_first_counts):
    """Time the NumPy oracle (a port of the reference's chain, oracle/chains.py) on ONE host core
    for about `budget_s` seconds on the same workload and the same (seed, realization) keying;
    also returns |SER_gpu - SER_oracle| on the realizations both sides computed."""
    from oracle import chains
    fn, kw = _oracle_chain(cfg)
    se = []
    t0 = time.perf_counter()
    r = 0
    while True:
        out = fn(chains.PhiloxRng(SEED, r), **kw)
        se.append(out["symbol_errors"])
        r += 1
        if time.perf_counter() - t0 >= budget_s or r >= len(gpu_first_counts):
            break
    dt = time.perf_counter() - t0
    n = len(se)
    ser_cpu = float(np.sum(se)) / (n * out["num_symbols"])
    ser_gpu = float(np.sum(gpu_first_counts[:n])) / (n * out["num_symbols"])
    return {"value": n / dt, "unit": "realizations/s", "cores": 1, "kind": "port",
            "sample": "%d realizations of the same workload, NumPy oracle (oracle/chains.py), 1 thread, %.1f s"
                      % (n, dt)}, abs(ser_gpu - ser_cpu), n


def _cpu_worker(job):
    """One process of the multi-core CPU leg: `n` realizations starting at `first` (NumPy oracle)."""
    cfg, first, n = job
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    from oracle import chains
    fn, kw = _oracle_chain(cfg)
    t0 = time.perf_counter()
    err = 0
    for r in range(first, first + n):
        err += fn(chains.PhiloxRng(SEED, r), **kw)["symbol_errors"]
    return n, time.perf_counter() - t0, err


def cpu_baseline_multicore(cfg, single_core_rate, budget_s):
    """The same oracle on every host core at once (one process per core, disjoint realization ranges),
    sized from the single-core rate to take about `budget_s` seconds."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    workers = max(1, min(cores // 2 if cores >= 4 else cores, 64))     # one per physical core, capped
    per = max(1, int(single_core_rate * budget_s * 0.7))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        done = pool.map(_cpu_worker, [(cfg, (1 << 30) + w * per, per) for w in range(workers)])
    wall = time.perf_counter() - t0
    n = sum(d[0] for d in done)
    busy = max(d[1] for d in done)
    return {"value": n / busy, "unit": "realizations/s", "cores": workers, "kind": "port",
            "sample": "%d realizations over %d processes (spawn), slowest worker %.1f s, wall %.1f s incl. start-up"
                      % (n, workers, busy, wall)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, "1")
    from pyphysim_amd.engine import Engine   # loads libmcle (shares torch's HIP runtime)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ        # any torch.distributed.run launch, even with 1 rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    eng = Engine(local_rank, args.dtype)
    run, units, workload = make_runner(eng, args.config, args.demod, args.dtype)
    batch = args.batch or {"c4": 65536, "c3": 131072, "c2": 4096, "c5": 262144, "f1": 98304, "f6": 131072}[args.config]

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    counters = eng.new_counters()
    # realization index space: warmup uses a disjoint range far away from the timed one
    for w in range(args.warmup):
        run((1 << 40) + (w * world + rank) * batch, batch, counters)
    if use_dist:        # bring the communicator up outside the timed region (same shape / dtype as the real exchange)
        warm = torch.zeros(6, dtype=torch.int64, device="cuda")
        dist.all_reduce(warm, op=dist.ReduceOp.SUM)
        warm2 = torch.zeros(2, dtype=torch.float64, device="cuda")
        dist.all_reduce(warm2, op=dist.ReduceOp.MAX)
    barrier()
    counters.zero()
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for s in range(args.steps):
        run((s * world + rank) * batch, batch, counters)
    kernel_ms = eng.timer_stop_ms()            # HIP events on the stream the kernels ran on
    local = eng.read_counters(counters)
    vec = torch.tensor([local[k] for k in ("n_realizations", "n_skipped", "sym_errors", "sym_errors_sq",
                                           "bit_errors", "bit_errors_sq")], dtype=torch.int64, device="cuda")
    if use_dist:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)     # RCCL over xGMI: the path's only exchange step
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(tmax[0]), float(tmax[1])
    tot = [int(v) for v in vec.tolist()]
    n_real = tot[0] + tot[1]
    assert n_real == args.steps * batch * world, (n_real, args.steps, batch, world)

    if rank == 0:
        value = n_real / elapsed
        per_launch_s = kernel_ms * 1e-3 / args.steps
        balg = B_ALG[args.config]
        achieved = balg * batch / per_launch_s / 1e9
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu = None        # what actually binds the fused kernels: VALU issue slots (last rocprofv3 PMC pass)
        ppath = os.path.join(REPO, "profiles", "r01", "%s_pmc_summary.json" % args.config)
        if os.path.exists(ppath):
            try:
                pm = json.load(open(ppath))
                waves_per_simd = 3
                valu = {"valu_busy_per_wave": pm["_derived"]["valu_active_per_wave"],
                        "waves_per_simd": waves_per_simd,
                        "valu_issue_frac": pm["_derived"]["valu_active_per_wave"] * waves_per_simd,
                        "source": "profiles/r01/%s_pmc_summary.json (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)" % args.config}
            except Exception:
                valu = None
        out = {
            "metric": "Monte Carlo realizations/sec (whole node) + SER abs-error vs ref",
            "value": value, "unit": "realizations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "realizations_per_step_per_gpu": batch, "demod": args.demod,
                       "symbols_per_realization": units, "parallelism": "realization-sharded x%d" % world,
                       "rng": "Philox4x32-10 keyed by (seed, realization)"},
            "ser": tot[2] / float(max(1, tot[0]) * units),
            "ber": tot[4] / float(max(1, tot[0]) * units * {"c2": 6, "c3": 2, "c4": 6, "c5": 4, "f1": 6, "f6": 2}[args.config]),
            "n_skipped": tot[1],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": {"c4": "k_run_mimo_ofdm", "c2": "k_run_flat", "c3": "k_run_ofdm_tdl_batch",
                                    "c5": "k_run_ia", "f1": "k_run_mimo_ofdm_tdl", "f6": "k_run_bd"}[args.config],
                         "kernel_ms_per_launch": per_launch_s * 1e3,
                         # SURVEY 8(d)'s reporting rule next to the tier's: HBM bytes the launch really moved / time
                         "hbm_measured_GBps": (traffic / per_launch_s / 1e9) if traffic is not None else None,
                         "valu": valu,
                         "algorithmic_bytes_per_realization": balg,
                         "traffic_source": "profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                         % args.config if traffic is not None else None,
                         "note": "fused kernel: achieved = staged-model algorithmic bytes / measured launch time; "
                                 "measured HBM traffic is far below it (data never leaves LDS); the kernel is "
                                 "VALU-issue bound, see DESIGN.md section 5.3"},
        }
        if world == 1 and not args.no_cpu:
            # per-realization counts of the first realizations for the SER cross-check
            res, se, be = eng_first_counts(eng, args, 16384 if args.config != "c2" else 128)
            cb, ser_err, n_chk = cpu_baseline(args.config, args.cpu_seconds, se)
            cb["host_cpu_count"] = os.cpu_count()
            cb["host_cpu_model"] = _cpu_model()
            out["cpu_baseline"] = cb
            out["ser_abs_err_vs_oracle"] = ser_err
            out["ser_check_realizations"] = n_chk
            out["speedup_vs_cpu_core"] = value / cb["value"]
            if args.cpu_multicore_seconds > 0:
                try:
                    mc = cpu_baseline_multicore(args.config, cb["value"], args.cpu_multicore_seconds)
                    out["cpu_baseline_all_cores"] = mc
                    out["speedup_vs_cpu_host"] = value / mc["value"]
                except Exception as exc:          # never let the optional leg break the bench line
                    out["cpu_baseline_all_cores"] = {"error": repr(exc)}
            if args.config == "c4" and args.demod == "slicer":
                # transparency: the same kernel with the exhaustive LDS-table demodulator
                run_md, _, _ = make_runner(eng, "c4", "mindist", args.dtype)
                cnt2 = eng.new_counters()
                run_md(1 << 41, batch, cnt2)
                eng.sync()
                t1 = time.perf_counter()
                for s2 in range(5):
                    run_md((1 << 41) + (s2 + 1) * batch, batch, cnt2)
                eng.sync()
                out["mindist_demod_realizations_per_s"] = 5 * batch / (time.perf_counter() - t1)
            if args.config == "c4" and not args.no_cpu:
                # the other workloads of SURVEY.md section 8 on the same device, 5 launches each (a second's work):
                # realizations/s and the kernel time of one launch, so that one bench line documents them all
                others = {}
                batches = {"c3": 131072, "c2": 4096, "c5": 262144, "f1": 98304, "f6": 131072}
                for cfg in ("c2", "c3", "c5", "f1", "f6"):
                    try:
                        run_o, units_o, wl_o = make_runner(eng, cfg, "slicer", args.dtype)
                        cnt_o = eng.new_counters()
                        run_o(1 << 42, batches[cfg], cnt_o)
                        eng.sync()
                        eng.timer_start()
                        for s2 in range(5):
                            run_o((1 << 42) + (s2 + 1) * batches[cfg], batches[cfg], cnt_o)
                        ms_o = eng.timer_stop_ms() / 5
                        c_o = eng.read_counters(cnt_o)
                        others[cfg] = {"workload": wl_o, "realizations_per_s": batches[cfg] / ms_o * 1e3,
                                       "kernel_ms_per_launch": ms_o, "realizations_per_launch": batches[cfg],
                                       "ser": c_o["sym_errors"] / float(max(1, c_o["n_realizations"]) * units_o)}
                    except Exception as exc:
                        others[cfg] = {"error": repr(exc)}
                out["other_workloads"] = others
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


def eng_first_counts(eng, args, n):
    from pyphysim_amd import _lib
    method = _lib.DEMOD_QAM_SLICER if args.demod == "slicer" else _lib.DEMOD_MINDIST
    nv = 1.0 / (10.0 ** (SNR_DB[args.config] / 10.0))
    if args.config == "c4":
        return eng.run_mimo_ofdm(4, 4, 1024, 16, 1024, 1, nv, SEED, 0, n, method=method, dtype=args.dtype,
                                 per_realization=True)
    if args.config == "c2":
        return eng.run_flat_fading(100000, nv, SEED, 0, n, method=method, dtype=args.dtype, per_realization=True)
    if args.config == "f1":
        from pyphysim_amd.channels import discretize_profile
        p_lin, d_idx = discretize_profile(np.array(F1_TAPS_DB), np.arange(5) * F1_TS, F1_TS)
        return eng.run_mimo_ofdm_tdl(4, 4, 1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, 0, n, Fd=10.0, Ts=F1_TS, L=8,
                                     method=method, dtype=args.dtype, per_realization=True)
    if args.config == "c5":
        return eng.run_ia(200, nv, SEED, 0, n, method=method, dtype=args.dtype, per_realization=True)[:3]
    if args.config == "f6":
        return eng.run_bd(3, 2, 500, 1.0, nv, SEED, 0, n, method=_lib.DEMOD_MINDIST, dtype=args.dtype,
                          per_realization=True)
    from pyphysim_amd.channels import discretize_profile
    Ts = 1.0 / (15e3 * 1024)
    p_lin, d_idx = discretize_profile(np.array([0.0, -3.0, -6.0, -9.0, -12.0]), np.arange(5) * Ts, Ts)
    return eng.run_ofdm_tdl(1024, 16, 1024, 1, nv, p_lin, d_idx, SEED, 0, n, Fd=10.0, Ts=Ts, L=8, dtype=args.dtype,
                            per_realization=True)


if __name__ == "__main__":
    main()
