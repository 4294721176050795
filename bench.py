#!/usr/bin/env python3
"""bench.py -- Monte Carlo realizations/s of the fused HIP link pipeline on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  * N == 1: runs in this process.
  * N > 1 and not already under a launcher (no RANK in the environment): bench.py starts its own N ranks,
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`,
    one rank per GPU over RCCL.  Launched by the driver's own torch.distributed.run it simply reads
    RANK / LOCAL_RANK / WORLD_SIZE.  Either way WORLD_SIZE must equal --gpus (asserted).

Workload (BASELINE.json north_star target, configs[3] geometry on ONE GPU per rank):
  4x4 MIMO (Blast, MMSE) + 64-QAM + OFDM-1024 (cp 16, all bins used), flat H ~ randn_c(4,4) per
  realization, SNR 25 dB; a "step" = one batch of --batch realizations per GPU through
  mcle_run_mimo_ofdm (data, channel and noise drawn on-chip from (seed, realization index)).
  Rank r owns the contiguous realization range [r*K*batch, (r+1)*K*batch) ("weak" scaling); the only
  exchange is ONE all-reduce (RCCL) of the 6-word integer counter vector, inside the timed region.

`--launch-check` exercises the launcher, the range split and the reduction WITHOUT a GPU (gloo, an integer
checksum per realization index instead of a kernel; no rate is reported) -- tests/test_bench_launch.py.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # before NumPy loads its BLAS:
    os.environ.setdefault(_v, "1")                                        # the CPU legs are per-core figures

import numpy as np  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

ROUND = "r06"
# algorithmic bytes per realization in the staged (operator-granular)
# model, complex64 samples / uint8 indices.
B_ALG = {"c2": 7_600_000, "c3": 211_360, "c4": 412_160, "c5": 62_000, "f1": 3_099_008,
         # f6 (6 streams x 500 symbols): gen+mod 27 000; precode, channel, filter 48 000 each; demod 27 000; count 6 000
         "f6": 204_000}
# Algorithmic floating-point operations per realization (complex MAC = 8, complex add = 2, 5 N log2 N per FFT),
# SURVEY.md section 8(d) "Flops per realization"; RNG and integer work is NOT counted (listed under "uncounted").
# Polynomial order of the tap model the TDL kernels run at the bench Doppler (host: truncation below 1e-8 in complex64, 1e-17 in
# complex128 -- csrc/pipeline_siso_tdl.hip, pipeline_mimo_tdl.hip): the order is a template argument of the dispatched kernel
TAP_ORDER = {"f32": 2, "f64": 5}
FLOPS = {
    "c4": {"H.T (4x4 x 1040 cMAC)": 133_120, "G.Y (4x4 x 1024 cMAC)": 131_072, "8 x FFT-1024": 409_600,
           "MMSE filter (f64)": 2_000, "slicer (4096 x 10)": 40_960},
    # config 3 / f1: what RUNS -- the rays of a symbol are folded once into order-K tap polynomials (k_tdl_symbol_polys), the link
    # kernel evaluates them by Horner (4 flops per order: a complex coefficient, a real abscissa) -- see flops_for(); until round 4
    # the config-3 model counted the literal sum of sinusoids per sample (249 600 flops) that no kernel executes any more
    "c3": {"2 x FFT-1024": 102_400, "TDL 5 taps x 1040 cMAC": 41_600,
           "equaliser (5-tap DFT + divide) x 1024": 55_296, "demod 1024 x 6": 6_144},
    "c2": {"Jakes 8 rays x 1e5 (phase, sincos, add)": 4_800_000, "fade + equalise 1e5 x 22": 2_200_000,
           "slicer 1e5 x 10": 1_000_000},
    "f1": {"8 x FFT-1024": 409_600, "TDL 5 taps x 16 links x 1040 cMAC": 665_600,
           "H(f) 1024 x 16 x 5 cMAC": 655_360,
           # per subcarrier: Hermitian Gram H^H H (10 entries x 4 cMAC = 320) + H^H y (16 cMAC = 128) + diagonal load (4) + complex
           # Cholesky 4 x 4 (~170 + 4 sqrt / 4 rcp) + two triangular substitutions (2 x 10 cMAC = 160) + slack = ~0.96 kflop
           # (rounds 3-5 booked ~1.1 kflop: 15 % generous)
           "MMSE solve per subcarrier 1024 x ~0.96 kflop": 983_040,
           "slicer": 40_960},
    "c5": {"link 600 x 15 cMAC": 72_000, "closed-form solve (f64)": 6_000, "demod 600 x 10": 6_000},
    "f6": {"link 3000 x (1 + 2) cMAC": 72_000, "BD solve + pinv (f64)": 20_000, "demod 3000 x 6": 18_000},
}
C3_LITERAL_JAKES_FLOPS = 249_600     # 5 taps x 8 rays x 1040 samples x (phase, sincos, add): the reference's evaluation, not the kernels'


def flops_for(cfg, dtype):
    """Algorithmic flops per realization of what the dispatched kernels execute (dtype picks the tap polynomials' order)."""
    f = dict(FLOPS[cfg])
    K = TAP_ORDER[dtype]
    if cfg == "c3":
        f["tap polynomials: 5 taps x 1040 samples x order-%d Horner (4 flops per order)" % K] = 5 * 1040 * 4 * K
        f["fold of 5 x 8 rays into %d coefficients each (phase, sincos, powers)" % (K + 1)] = 40 * (12 + 4 * (K + 1))
    if cfg == "f1":
        f["tap polynomials: 80 links x 1040 samples x order-%d Horner (4 flops per order)" % K] = 80 * 1040 * 4 * K
        f["fold of 80 x 8 rays into %d coefficients each" % (K + 1)] = 640 * (12 + 4 * (K + 1))
    return f


UNCOUNTED = {"c4": "Philox4x32-10: ~2 350 blocks (4 176 CN samples + 4 096 symbol bytes) = ~140 k integer ops; "
                   "Box-Muller: 4 176 x (log, sqrt, sin, cos) transcendental ops"}
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY_GBPS = 6290.0      # same guide: measured float4 copy (a reference figure; the line quotes what THIS box measures)
MEASURED = {"copy_GBps": None, "stream": None}   # this box's achievable HBM rate measured in THIS run by the library's own
                                                 # streaming kernels (scripts/bench_staged_c4.measure_hbm_stream), rank 0
FP32_PEAK_TFLOPS = 157.3    # same guide: FP32 vector peak = FP32-input MFMA peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz)
# FP64: 16 lanes x 1 FMA per clock and SIMD = half the guide's FP32 figure (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz); the guide
# does not list it, so it was measured (scripts/experiments/f64_rates.hip -> profiles/r03/f64_rates.txt): v_mfma_f64_16x16x4_f64
# 77.6 TFLOP/s, v_fma_f64 65.9 TFLOP/s, and the two do not overlap on a SIMD.
FP64_PEAK_TFLOPS = 78.6
FP64_MEASURED = {"v_mfma_f64_16x16x4_f64": 77.6, "v_fma_f64": 65.9, "source": "profiles/r03/f64_rates.txt"}
PEAK_TFLOPS = {"f32": FP32_PEAK_TFLOPS, "f64": FP64_PEAK_TFLOPS}
KERNEL = {"c4": "k_run_mimo_ofdm_planar", "c2": "k_run_flat_mfma", "c3": "k_run_ofdm_tdl_wave", "c5": "k_ia_link",
          "f1": "k_run_mimo_ofdm_tdl_wave", "f6": "k_bd_link"}
KERNEL_F64 = {"c4": "k_run_mimo_ofdm_planar", "c2": "k_run_flat", "c3": "k_run_ofdm_tdl_wave", "c5": "k_ia_link",
              "f1": "k_run_mimo_ofdm_tdl_wave", "f6": "k_bd_link"}


ACTIVE_OPTS = {}            # --opt name=value of this run (main fills it): some options change which kernel a configuration runs


def kernel_name(cfg, dtype):
    if cfg == "c4" and dtype == "f32" and ACTIVE_OPTS.get("f32_mfma"):
        return "k_run_mimo_ofdm_mfma"                 # the matrix-core kernel of rounds 2-3 (option f32_mfma = 1)
    if cfg == "c4" and ACTIVE_OPTS.get("f64_generic") or (cfg == "c4" and dtype == "f32" and ACTIVE_OPTS.get("no_mfma")):
        return "k_run_mimo_ofdm<"                     # the generic radix-4 kernel
    if cfg == "c3" and (ACTIVE_OPTS.get("tdl_kernel") == 1 or ACTIVE_OPTS.get("no_mfma")):
        return "k_run_ofdm_tdl_batch" if (dtype == "f64" or ACTIVE_OPTS.get("no_mfma")) else "k_run_ofdm_tdl_mfma"
    if cfg == "f1" and ACTIVE_OPTS.get("mimo_tdl_kernel") == 1:
        return "k_run_mimo_ofdm_tdl"                  # the workgroup-cooperative kernel of rounds 1-4
    if cfg == "c4" and dtype == "f64" and ACTIVE_OPTS.get("f64_threads", 0) in (0, 263, 264):
        return "k_run_mimo_ofdm_pw"                   # quarter-wave decomposition, channel AND decode on the matrix cores (round 6, default)
    if cfg == "c4" and dtype == "f64" and ACTIVE_OPTS.get("f64_threads", 0) in (260, 262):
        return "k_run_mimo_ofdm_qw"                   # the first quarter-wave kernel (VALU decode)
    if cfg in ("c5", "f6") and not ACTIVE_OPTS.get("walk_legacy"):
        return "k_link_walk<"                         # the packed walk (round 6, csrc/walk_f64.hpp), either arithmetic
    return (KERNEL_F64 if dtype == "f64" else KERNEL)[cfg]

KERNEL_NOTE = {   # configs whose step is two launches: the time spans both, the counters are the named (dominant) kernel's
    "c4": "a step = per slice of 2^18 realizations k_mimo_filters_planar<float> (channel draw in float, receive filter in f64, one "
          "thread per realization, ~1 % of the time) + k_run_mimo_ofdm_planar<float> (radix-16 register passes, one transform per "
          "wavefront; since round 4 the default ahead of the matrix-core kernel k_run_mimo_ofdm_mfma, option f32_mfma = 1); "
          "kernel_ms_per_launch spans them",
    ("c4", "f64"): "a step = per slice of 2^18 realizations k_mimo_filters_planar (channel draw + f64 receive filter, one thread per "
                   "realization, ~1 % of the time) + k_run_mimo_ofdm_pw<4> (round 6: a wavefront owns one time class n mod 4 of all four "
                   "antennas, samples in registers between radix-16 passes, three workgroups per CU, the channel contraction AND the "
                   "Blast decode on v_mfma_f64_4x4x4; option f64_threads=260: k_run_mimo_ofdm_qw, the same with the decode on the VALU; "
                   "261: the planar kernel k_run_mimo_ofdm_planar of rounds 3-5); kernel_ms_per_launch spans them",
    ("c3", "f64"): "a step = k_tdl_symbol_polys<double> (fading records) + k_run_ofdm_tdl_wave<double> (one realization per "
                   "wavefront; option tdl_kernel=1: k_run_ofdm_tdl_batch<double, 1024, 2>) per slice of <= 2 GiB of records; "
                   "kernel_ms_per_launch spans them",
    "f1": "a step = k_mimo_tdl_symbol_polys<T, true> (the symbols' fading records, one thread per fading process) + "
          "k_run_mimo_ofdm_tdl_wave (one receive antenna per wavefront, default since round 5; option mimo_tdl_kernel=1: the "
          "workgroup-cooperative k_run_mimo_ofdm_tdl) per slice of <= 4 GiB of records; kernel_ms_per_launch spans them",
    "c3": "a step = k_tdl_symbol_polys (fading records) + k_run_ofdm_tdl_wave<float> (one realization per wavefront, default since "
          "round 4; option tdl_kernel=1: the matrix-core kernel k_run_ofdm_tdl_mfma) per slice of <= 2 GiB of records; "
          "kernel_ms_per_launch spans them",
    "c5": "a step = k_ia_solve_links (per-lane solve, ~18 % of the time) + k_ia_link (symbol walk); kernel_ms_per_launch spans both",
    ("c5", "f64"): "a step = k_ia_solve_links<double> (per-lane solve) + k_link_walk<double, IaWalk, decision form> (round 6: the lane pairs "
                   "of 16 realizations as one index space, csrc/walk_f64.hpp; option walk_legacy=1: k_ia_link<double>); "
                   "kernel_ms_per_launch spans both",
    ("f6", "f64"): "a step = k_bd_solve_links_static<double> (per-lane solve) + k_link_walk<double, BdWalk<3, 2>, decision form> (round 6, "
                   "csrc/walk_f64.hpp; option walk_legacy=1: k_bd_link<double>); kernel_ms_per_launch spans both",
    "f6": "a step = k_bd_solve_links (per-lane solve) + k_bd_link (symbol walk); kernel_ms_per_launch spans both"}
# realizations per GPU and step: sized so that a step is >= 15 ms on the fastest kernel of the configuration -- K = 20 steps
# then time >= 0.3 s, long enough that one rank's scheduling hiccup of a millisecond is < 0.5 % of an 8-rank run's region
# (round 2 timed 29 ms in all).  Nothing in BASELINE.json fixes the batch; a step is one call of the pipeline's C entry point.
BATCH = {"c4": 1048576, "c3": 2097152, "c2": 131072, "c5": 4194304, "f1": 393216, "f6": 1048576}
BATCH_SURVEY = {"c4": 65536, "c3": 131072, "c2": 16384, "c5": 262144, "f1": 98304, "f6": 131072}   # other_workloads legs


def profile_specs():
    """tag -> the bench arguments (config, dtype, demodulator, batch, options) a committed profile of that tag is taken with, and
    the dominant kernel's name.  ONE table for scripts/prof_r06.sh and scripts/collect_profiles.py: every `other_workloads` leg is
    profiled at the batch and the demodulator its leg prints (round 5 profiled config 2 with another
    demodulator and four configurations at another batch than their legs), the headline at the size of one dispatch of a step
    (the pipeline cuts a step into slices of 2^18 realizations).  tests/test_bench_launch.py holds the table to BATCH_SURVEY."""
    specs = {}
    hl = min(BATCH["c4"], 1 << 18)

    def add(tag, cfg, dtype, demod, batch, opts=(), leg=None, note=None):
        saved = dict(ACTIVE_OPTS)
        ACTIVE_OPTS.clear()
        for o in opts:
            k, _, v = o.partition("=")
            ACTIVE_OPTS[k] = int(v)
        try:
            needle = kernel_name(cfg, dtype)
        finally:
            ACTIVE_OPTS.clear()
            ACTIVE_OPTS.update(saved)
        specs[tag] = {"config": cfg, "dtype": dtype, "demod": demod, "batch": batch, "opts": list(opts), "kernel": needle,
                      "leg": leg, "note": note}
    add("c4_f64", "c4", "f64", "mindist", hl, leg="value", note="the headline: quarter-wave decomposition with channel and decode on the matrix cores, min-distance (certificate)")
    add("c4_f64_qw", "c4", "f64", "mindist", hl, opts=("f64_threads=260",), note="the first quarter-wave kernel, VALU decode (A/B)")
    add("c4_f64sl", "c4", "f64", "slicer", hl, leg="rates.f64.slicer")
    add("c4_f64_planar", "c4", "f64", "mindist", hl, opts=("f64_threads=261",), note="the planar kernel of rounds 3-5 (A/B)")
    add("c4", "c4", "f32", "slicer", hl, leg="rates.f32.slicer")
    add("c4md", "c4", "f32", "mindist", hl, leg="rates.f32.mindist")
    add("c4md_mfma", "c4", "f32", "mindist", BATCH_SURVEY["c4"], opts=("f32_mfma=1",), leg="other_workloads.c4_f32_mfma")
    for cfg in ("c2", "c3", "c5", "f1", "f6"):
        for dt in ("f32", "f64"):
            add(cfg + ("_f64" if dt == "f64" else ""), cfg, dt, "slicer", BATCH_SURVEY[cfg], leg="other_workloads.%s.%s" % (cfg, dt))
    add("c3_mfma", "c3", "f32", "slicer", BATCH_SURVEY["c3"], opts=("tdl_kernel=1",), note="the matrix-core kernel of rounds 2-3 (A/B)")
    add("f1_coop", "f1", "f32", "slicer", BATCH_SURVEY["f1"], opts=("mimo_tdl_kernel=1",), note="the cooperative kernel of rounds 1-4 (A/B)")
    return specs


BITS = {"c2": 6, "c3": 2, "c4": 6, "c5": 4, "f1": 6, "f6": 2}
# BASELINE.json's literal realization counts: config 4 "1e6 realizations sharded over 8", config 5 "1e5 realizations on 8" (configs
# 2 and 3: 1e6 on one GPU) -- what the `strong` block of a multi-rank line splits over the ranks
STRONG_TOTAL = {"c4": 10 ** 6, "c5": 10 ** 5, "c2": 10 ** 6, "c3": 10 ** 6, "f1": 10 ** 5, "f6": 10 ** 5}
SEED = 20260927
SNR_DB = {"c1": 10.0, "c2": 20.0, "c3": 20.0, "c4": 25.0, "c5": 20.0, "f1": 25.0, "f6": 15.0}
F1_TS = 1.0 / (15e3 * 1024)
F1_TAPS_DB = (0.0, -3.0, -6.0, -9.0, -12.0)
COUNTER_KEYS = ("n_realizations", "n_skipped", "sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq")
PMC_CHILD_STEPS = 4         # steps of a counter child run: --warmup 1 --steps 3, no pre-roll
PMC_PASSES = (   # one rocprofv3 --pmc run each (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md)
    ("sq", "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 "
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"),
    ("fetch", "FETCH_SIZE"),
    ("write", "WRITE_SIZE"),
    # the VALU instruction MIX, for an issue-time fraction that cannot exceed 1 (4 x SQ_ACTIVE_INST_VALU / cycles
    # books four cycles per instruction and read 1.04 - 1.11 for streams of 2.7-cycle f32 / integer instructions)
    ("mix", "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 "
            "SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"),
    ("mix2", "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_F32"),
)
# Cheapest MEASURED issue cost of a wave instruction per class, cycles per SIMD at >= 2 wavefronts per SIMD
# (scripts/experiments/f64_rates.hip -> profiles/r03/f64_rates.txt, f32_rates.hip -> profiles/r04/f32_rates.txt): a LOWER bound of
# the time the SIMDs spend issuing a counted instruction stream, so sum(count x cost) / SIMD-cycles cannot exceed 1
ISSUE_COST = {"SQ_INSTS_VALU_ADD_F64": 4.8, "SQ_INSTS_VALU_MUL_F64": 4.8, "SQ_INSTS_VALU_FMA_F64": 4.4, "SQ_INSTS_VALU_TRANS_F64": 16.4,
              "SQ_INSTS_VALU_ADD_F32": 2.6, "SQ_INSTS_VALU_MUL_F32": 2.6, "SQ_INSTS_VALU_FMA_F32": 2.6, "SQ_INSTS_VALU_TRANS_F32": 8.3,
              "SQ_INSTS_VALU_INT32": 2.66, "SQ_INSTS_VALU_INT64": 4.3, "SQ_INSTS_VALU_CVT": 4.2, "_other": 2.66}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preroll-ms", type=float, default=250.0,
                    help="untimed launches (a disjoint index range) before the W warm-up steps until this much wall time "
                         "has passed: a fresh GPU needs ~0.2 s of work to reach its clocks (first launches of a run were "
                         "measured 7 %% slower); 0 disables")
    ap.add_argument("--profile-spec", default=None, metavar="TAG",
                    help="print the bench arguments of profile tag TAG (or 'list': every tag) and exit: scripts/prof_r06.sh")
    ap.add_argument("--config", default="c4", choices=["c2", "c3", "c4", "c5", "f1", "f6"])
    ap.add_argument("--batch", type=int, default=0, help="realizations per GPU per step")
    ap.add_argument("--single-demod", action="store_true",
                    help="time only the --demod / --dtype combination (profiling runs: every launch is then the same kernel work)")
    ap.add_argument("--demod", default="mindist", choices=["slicer", "mindist"],
                    help="demodulator `value` is quoted on (mindist = the north star's min-distance search over the LDS "
                         "constellation table); the c4 line carries the rate of both")
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"],
                    help="arithmetic `value` is quoted on: f64 = complex128, the reference's own precision (default); the "
                         "line carries the complex64 rate of the same workload next to it (dtype_rates)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--cpu-multicore-seconds", type=float, default=6.0,
                    help="budget of the all-cores CPU leg (0 disables it)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU legs and the other-workload survey")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="collect the VALU / MFMA / HBM counters of the dominant kernel with rocprofv3 child runs "
                         "(auto: single rank, rocprofv3 on PATH, CPU legs not disabled)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-selection option of the context (mcle_ctx_set_option; names: pyphysim_amd._lib.OPTIONS), "
                         "e.g. --opt no_mfma=1 --opt grid_oversub=4; repeatable, for A/B runs")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of a multi-rank run: nccl (= RCCL, the default and what the driver gets); "
                         "gloo exists so that the N > 1 code path can be executed on a box with fewer GPUs than ranks")
    ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                    help="who carries the counter all-reduce of a multi-rank run: native = the product's own exchange, "
                         "mcle_counters_allreduce (csrc/comm.hip: device-resident counters, one grouped RCCL all-reduce on the "
                         "context stream; communicator from pyphysim_amd.distributed.NativeComm); torch = read_counters -> "
                         "torch.distributed.all_reduce; auto = native on the nccl backend when its communicator comes up on "
                         "every rank, else torch.  The line says which ran (rccl.exchange_impl)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="rank r uses GPU r mod (visible GPUs) instead of failing when there are fewer GPUs than ranks "
                         "(tests of the N > 1 path on one GPU, together with --dist-backend gloo; never a scaling figure)")
    ap.add_argument("--strong-total", type=int, default=0,
                    help="total realizations of the strong-scaling leg of a multi-rank run (0: BASELINE.json's literal count "
                         "for the configuration: config 4 1e6, config 5 1e5); split contiguously over the ranks")
    ap.add_argument("--strong-reps", type=int, default=5, help="timed repetitions of the strong-scaling leg (median reported)")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU: run the rank launcher, the range split and the reduction on gloo with an integer "
                         "checksum per realization index instead of a kernel (prints ranges + counters, no rate)")
    return ap.parse_args()


# ---- launcher ---------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under torch.distributed.run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    # (the port is free when it is picked, not necessarily when the launcher binds it: a launch that dies within seconds -- the
    #  rendezvous' EADDRINUSE, before any rank has printed anything -- is repeated on another port, twice at most)
    rc = 1
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.time()
        rc = subprocess.call(cmd, env=env)
        if rc == 0 or time.time() - t0 > 20.0:
            break
        print("bench.py: launcher exited with %d after %.1f s, trying another rendezvous port" % (rc, time.time() - t0), file=sys.stderr)
    return rc


# ---- workloads --------------------------------------------------------------------------------------------------
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


def _physical_cores():
    """(physical cores, logical CPUs) of the host: distinct (physical id, core id) pairs of /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        if pairs:
            return min(len(pairs), logical), logical
    except OSError:
        pass
    return max(1, logical // 2 if logical >= 4 else logical), logical


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
    """One process of the multi-core CPU leg: realizations from `first` on for `seconds` of wall time (NumPy oracle; a TIME
    budget, not a count: with every core busy a process runs several times slower than the one-core leg suggests)."""
    cfg, first, seconds = job
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"
    from oracle import chains
    fn, kw = _oracle_chain(cfg)
    fn(chains.PhiloxRng(SEED, first), **kw)                 # imports / first-call set-up outside the clock
    t0 = time.perf_counter()
    err, n = 0, 0
    while time.perf_counter() - t0 < seconds:
        err += fn(chains.PhiloxRng(SEED, first + 1 + n), **kw)["symbol_errors"]
        n += 1
    return n, time.perf_counter() - t0, err


def cpu_baseline_multicore(cfg, single_core_rate, budget_s):
    """The same oracle on every host core at once (one process per core, disjoint realization ranges),
    sized from the single-core rate to take about `budget_s` seconds."""
    import multiprocessing as mp
    physical, logical = _physical_cores()
    try:
        physical = min(physical, len(os.sched_getaffinity(0)))        # what this process may actually use
    except (AttributeError, OSError):
        pass
    workers = max(1, min(physical, 256))                              # one process per physical core
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        done = pool.map(_cpu_worker, [(cfg, (1 << 30) + w * (1 << 20), budget_s) for w in range(workers)], chunksize=1)
    wall = time.perf_counter() - t0
    n = sum(d[0] for d in done)
    busy = max(d[1] for d in done)
    return {"value": n / busy, "unit": "realizations/s", "cores": workers, "kind": "port",
            "host_physical_cores": physical, "host_logical_cpus": logical,
            "scaling_vs_one_core": (n / busy) / single_core_rate,
            "sample": "%d realizations over %d processes (spawn) = one per physical core of %d (%d logical CPUs), each for a "
                      "fixed %.1f s of wall time, wall %.1f s incl. start-up; scaling against the one-core leg %.1f x (shared "
                      "memory bandwidth / boost clocks)"
                      % (n, workers, physical, logical, busy, wall, (n / busy) / single_core_rate)}


# ---- rocprofv3 counters of the dominant kernel, collected by child runs of this file ----------------------------
def _parse_pmc_csv(folder, needle):
    """-> {counter: SUM over every dispatch of the kernel in the run}, plus "_dispatches" (per counter pass the same number)."""
    import csv
    import glob
    agg, n = {}, {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if needle + "<" in row["Kernel_Name"] or needle + "(" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] = agg.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                n[row["Counter_Name"]] = n.get(row["Counter_Name"], 0) + 1
    if n:
        agg["_dispatches"] = max(n.values())
    return agg


def derive_pmc(c, per_launch):
    """Counter SUMS over a run's dispatches of the kernel (or means per launch) -> the fractions the bench line quotes.  Shared
    with scripts/collect_profiles.py.  `per_launch` = the realizations those counters cover: with sums, every realization of the
    profiled run (steps x batch) -- a pipeline that cuts a step into several dispatches of unequal size (the record-buffer
    slices of the TDL kernels) is then normalised correctly, which means per launch were not.
    SQ_* cycle counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles (MI355X_MICROARCH.md, constants
    table); GRBM_GUI_ACTIVE sums the 8 XCDs; the chip has 1024 SIMDs.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE)
    KiB: on gfx950 FETCH_SIZE reports half of a coalesced stream's bytes (same guide, HBM section)."""
    d = {}
    g = c.get
    if g("GRBM_GUI_ACTIVE"):
        simd_cycles = g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0
        if g("SQ_ACTIVE_INST_VALU") is not None:
            d["valu_busy_chip"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / simd_cycles
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["mfma_busy_chip"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles
    # issue-cost-weighted VALU fraction: every counted class at its cheapest measured issue cost, the uncounted rest (moves, selects,
    # lane swaps, packed forms) at the cheapest 32-bit cost; matrix-core instructions excluded (mfma_busy_chip is their share)
    if g("GRBM_GUI_ACTIVE") and g("SQ_INSTS_VALU") is not None and all(g(k) is not None for k in ISSUE_COST if k != "_other"):
        typed = sum(g(k) for k in ISSUE_COST if k != "_other")
        mfma_n = (g("SQ_INSTS_VALU_MFMA_F64") or 0.0) + (g("SQ_INSTS_VALU_MFMA_F32") or 0.0) or (g("SQ_INSTS_MFMA") or 0.0)
        other = max(0.0, g("SQ_INSTS_VALU") - typed - mfma_n)
        cyc = sum(g(k) * c for k, c in ISSUE_COST.items() if k != "_other") + other * ISSUE_COST["_other"]
        d["valu_issue_frac"] = cyc / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)
        d["valu_mix_per_realization"] = dict({k.replace("SQ_INSTS_VALU_", "").lower(): g(k) / per_launch for k in ISSUE_COST if k != "_other"},
                                             other=other / per_launch, mfma=mfma_n / per_launch)
    if g("SQ_ACTIVE_INST_VALU") is not None and g("SQ_WAVE_CYCLES"):
        d["valu_active_per_wave"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAIT_INST_ANY") is not None and g("SQ_WAVE_CYCLES"):
        d["wait_inst_any_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_INSTS_VALU") is not None:
        d["valu_wave_insts_per_realization"] = g("SQ_INSTS_VALU") / per_launch
    if g("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None:
        d["mfma_f32_mops_per_realization"] = g("SQ_INSTS_VALU_MFMA_MOPS_F32") / per_launch
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["hbm_bytes_per_launch"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
        d["hbm_bytes_per_realization"] = d["hbm_bytes_per_launch"] / per_launch
    return d


def collect_pmc_live(args, batch, dtype, demod):
    """Three `rocprofv3 --pmc` child runs of this bench (1 warm-up + 3 timed steps of `batch` realizations each) -> counter SUMS
    over every dispatch of the dominant kernel: they cover PMC_CHILD_STEPS x batch realizations."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    counters = {}
    root = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for tag, names in PMC_PASSES:
        out_dir = os.path.join(root, tag)
        cmd = [exe, "--pmc"] + names.split() + ["--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
                                                 sys.executable, os.path.abspath(__file__), "--config", args.config,
                                                 "--demod", demod, "--dtype", dtype, "--batch", str(batch),
                                                 "--steps", "3", "--warmup", "1", "--no-cpu", "--pmc", "off", "--single-demod", "--preroll-ms", "0"]
        for item in args.opt:
            cmd += ["--opt", item]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=180, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = _parse_pmc_csv(out_dir, kernel_name(args.config, dtype))
            counters["_dispatches_" + tag] = got.pop("_dispatches", 0)
            counters.update(got)
        except Exception as exc:       # a failed pass must not break the bench line
            counters["_error_" + tag] = repr(exc)
    shutil.rmtree(root, ignore_errors=True)
    if not any(not k.startswith("_") for k in counters):
        return None, "no counter rows for %s" % kernel_name(args.config, dtype)
    return counters, None


def committed_pmc(cfg, dtype):
    """-> (counter means per launch, path, realizations per launch of that profile)"""
    path = os.path.join(REPO, "profiles", ROUND, "%s%s_pmc_summary.json" % (cfg, "_f64" if dtype == "f64" else ""))
    if not os.path.exists(path):
        return None, None, None
    try:
        doc = json.load(open(path))
        if "_realizations_total" in doc:       # round 5 summaries: sums over every dispatch, the realizations they cover
            return ({k: v["sum"] for k, v in doc.items() if not k.startswith("_")}, os.path.relpath(path, REPO),
                    int(doc["_realizations_total"]))
        return ({k: v["mean_per_launch"] for k, v in doc.items() if not k.startswith("_")}, os.path.relpath(path, REPO),
                int(doc.get("_realizations_per_launch", BATCH_SURVEY[cfg])))
    except Exception:
        return None, None, None


def roofline_block(args, dtype, batch, per_launch_s, rate_kernel, d, pmc_source):
    """What binds the dominant kernel, as fractions in (0, 1]:
      frac            = algorithmic flops per realization x realizations/s of the kernel / peak of the dtype's datapath
      hbm.frac        = min(B_alg, measured HBM bytes) per realization x rate / 8 TB/s   (SURVEY 8(d)'s rule)
      valu_busy_chip  = VALU-active SIMD-cycles / all SIMD-cycles (rocprofv3 counters)"""
    flops = flops_for(args.config, dtype)
    f_total = float(sum(flops.values()))
    achieved_tf = f_total * rate_kernel / 1e12
    balg = B_ALG[args.config] * (2 if dtype == "f64" else 1)       # complex128 samples: twice the bytes of the staged model
    peak = PEAK_TFLOPS[dtype]
    d = d or {}                                   # counters derived per realization (derive_pmc), possibly from a smaller launch
    measured = d.get("hbm_bytes_per_realization")
    hbm = {"b_alg_bytes_per_realization": balg, "measured_bytes_per_realization": measured,
           "b_alg_over_measured": (balg / measured) if measured else None}
    if measured is not None:
        eff = min(float(balg), measured)
        hbm.update(achieved_GBps=eff * rate_kernel / 1e9, frac=eff * rate_kernel / 1e9 / HBM_PEAK_GBPS,
                   frac_of_achievable_hbm=eff * rate_kernel / 1e9 / (MEASURED["copy_GBps"] or HBM_COPY_GBPS),
                   copy_GBps=MEASURED["copy_GBps"] or HBM_COPY_GBPS,
                   copy_GBps_source="best of the library's copy / read / triad / write kernels over 1 GiB, measured in this run "
                                    "(mcle_hbm_stream_rate)" if MEASURED["copy_GBps"] else
                                    "MI355X_MICROARCH.md constant (not measured in this run)",
                   peak_GBps=HBM_PEAK_GBPS,
                   rule="min(B_alg, measured bytes) x rate (SURVEY.md 8(d)); measured = (2*FETCH_SIZE + WRITE_SIZE) KiB")
    block = {"bound": "valu", "dtype": dtype, "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
             "frac": achieved_tf / peak,
             "peak_note": ("FP64 vector = matrix peak, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz; measured on this chip: %s"
                           % json.dumps(FP64_MEASURED)) if dtype == "f64" else
                          "FP32 vector = f32-input MFMA peak (MI355X_MICROARCH.md)",
             "traffic": (measured * batch) if measured is not None else None,      # HBM bytes per launch of `batch` realizations
             "kernel": kernel_name(args.config, dtype), "kernel_ms_per_launch": per_launch_s * 1e3,
             "kernel_note": KERNEL_NOTE.get((args.config, dtype), KERNEL_NOTE.get(args.config)),
             "realizations_per_launch": batch,
             "flops_per_realization": f_total, "flops_breakdown": flops, "uncounted": UNCOUNTED.get(args.config),
             "flops_model": "what the dispatched kernels execute (tap polynomials of order %d in this arithmetic)" % TAP_ORDER[dtype]
             if args.config in ("c3", "f1") else "SURVEY.md section 8(d)",
             "frac_literal_jakes_model": ((f_total + C3_LITERAL_JAKES_FLOPS) * rate_kernel / 1e12 / peak) if args.config == "c3" else None,
             # the SAME rate on SURVEY-style flops only (FLOPS[cfg]: transforms, delay line, H(f), solves, decisions -- without the
             # implementation-sized tap polynomials and ray folds flops_for() adds): VERDICT r05 weak 6
             "frac_survey_flops_model": (sum(FLOPS[args.config].values()) * rate_kernel / 1e12 / peak)
             if args.config in ("c3", "f1") else None,
             "hbm": hbm,
             # valu_busy_chip: the issue-cost-weighted fraction where the instruction mix was collected (cannot exceed 1); the raw
             # counter ratio 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles is kept next to it under its own name -- it is NOT a fraction for
             # streams of sub-4-cycle instructions
             "valu_busy_chip": d.get("valu_issue_frac") if d.get("valu_issue_frac") is not None
             else (min(1.0, d["valu_busy_chip"]) if d.get("valu_busy_chip") is not None else None),
             "valu_busy_definition": "sum over instruction classes of count x cheapest measured issue cost / SIMD-cycles (bench.py "
                                     "ISSUE_COST)" if d.get("valu_issue_frac") is not None else
                                     "min(1, 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles): instruction mix not collected",
             "valu_active_x4_over_simd_cycles": d.get("valu_busy_chip"),
             "valu_mix_per_realization": d.get("valu_mix_per_realization"),
             "mfma_busy_chip": d.get("mfma_busy_chip"),
             # f32-input MFMA and VALU instructions share one FP32 datapath per SIMD on gfx950 (measured:
             # scripts/experiments/mfma_valu_overlap.hip, profiles/r02/mfma_valu_overlap.txt), so the two add up
             "fp32_datapath_busy_chip": ((d.get("valu_issue_frac") if d.get("valu_issue_frac") is not None else min(1.0, d["valu_busy_chip"]))
                                         + d["mfma_busy_chip"])
             if dtype == "f32" and d.get("valu_busy_chip") is not None and d.get("mfma_busy_chip") is not None else None,
             "valu_busy_note": "4 x SQ_ACTIVE_INST_VALU / SIMD-cycles: the counter books 4 cycles per wave instruction, while v_add / "
                               "v_mul / v_fma_f32 and the 32-bit logic ops issue in 2.7 - 3.0 cycles at >= 2 wavefronts per SIMD "
                               "(scripts/experiments/f32_rates.hip -> profiles/r04/f32_rates.txt), so a kernel made of them reads "
                               "above 1: >= 1 means the VALU issue is saturated" if dtype == "f32" else None,
             "valu_wave_insts_per_realization": d.get("valu_wave_insts_per_realization"),
             "mfma_f32_mops_per_realization": d.get("mfma_f32_mops_per_realization"),
             "wait_inst_any_frac": d.get("wait_inst_any_frac"),
             "dispatches_per_step": d.get("dispatches_per_step"),
             "counters_source": pmc_source,
             "note": "fused kernel: every intermediate of a realization lives in LDS / registers, so HBM traffic is "
                     "B_alg / %s of the staged model and the kernel is bound by the SIMDs' %s datapath%s; "
                     "frac = algorithmic flops (RNG excluded) / peak"
                     % (("%.0f" % (balg / measured)) if measured else "?", "FP64" if dtype == "f64" else "FP32",
                        " (VALU, plus v_mfma_f64_4x4x4 for the H x and G y contractions of the part-wave kernel -- mfma_busy_chip; the "
                        "f64 MFMA forms are no denser than v_fma_f64 and do not overlap with it, DESIGN.md sections 5.5 / 5.10)"
                        if dtype == "f64" else
                        " (VALU; the matrix-core form of this configuration, option f32_mfma, shares that datapath: f32 MFMA and "
                        "VALU instructions do not overlap on gfx950)")}
    return block


# ---- launcher self-test (no GPU) --------------------------------------------------------------------------------
def _checksum_counts(first, count):
    """Integer stand-in for a kernel launch in --launch-check: a pure function of the realization index."""
    idx = np.arange(first, first + count, dtype=np.uint64)
    h = (idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(52)             # 12-bit pseudo error count
    se = h.astype(np.int64)
    be = (h >> np.uint64(3)).astype(np.int64)
    return [int(count), 0, int(se.sum()), int((se * se).sum()), int(be.sum()), int((be * be).sum())]


def launch_check(args, rank, world):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = args.batch or 1024
    lo = rank * args.steps * batch
    tot = np.zeros(6, dtype=np.int64)
    for s in range(args.steps):
        tot += np.array(_checksum_counts(lo + s * batch, batch), dtype=np.int64)
    vec = torch.tensor(tot)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    ranges = [None] * world
    dist.all_gather_object(ranges, [int(lo), int(lo + args.steps * batch)])
    # the strong split of main(): `total` realizations (BASELINE's literal count) cut contiguously, in calls of <= batch
    s_total = args.strong_total or STRONG_TOTAL[args.config]
    s_lo, s_hi = (s_total * rank) // world, (s_total * (rank + 1)) // world
    s_tot = np.zeros(6, dtype=np.int64)
    pos = s_lo
    while pos < s_hi:
        n = min(batch, s_hi - pos)
        s_tot += np.array(_checksum_counts((17 << 36) + pos, n), dtype=np.int64)
        pos += n
    s_vec = torch.tensor(s_tot)
    dist.all_reduce(s_vec, op=dist.ReduceOp.SUM)
    s_ranges = [None] * world
    dist.all_gather_object(s_ranges, [int(s_lo), int(s_hi)])
    # ... and of BASELINE.json's literal counts of the two multi-GPU configurations (config 4: 10^6, config 5: 10^5), whatever
    # --config says: what a multi-rank bench line reports as `strong` and `strong_c5`
    literal = {}
    for cfg_l in ("c4", "c5"):
        t_l = STRONG_TOTAL[cfg_l]
        lo_l, hi_l = (t_l * rank) // world, (t_l * (rank + 1)) // world
        acc = np.zeros(6, dtype=np.int64)
        pos = lo_l
        while pos < hi_l:
            n = min(1 << 16, hi_l - pos)
            acc += np.array(_checksum_counts((19 << 36) + pos, n), dtype=np.int64)
            pos += n
        v_l = torch.tensor(acc)
        dist.all_reduce(v_l, op=dist.ReduceOp.SUM)
        r_l = [None] * world
        dist.all_gather_object(r_l, [int(lo_l), int(hi_l)])
        literal[cfg_l] = {"total_realizations": t_l, "rank_ranges": r_l,
                          "counters": dict(zip(COUNTER_KEYS, [int(v) for v in v_l.tolist()]))}
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "steps": args.steps, "value": None,
                          "rank_ranges": ranges, "counters": dict(zip(COUNTER_KEYS, [int(v) for v in vec.tolist()])),
                          "strong": {"total_realizations": s_total, "rank_ranges": s_ranges,
                                     "counters": dict(zip(COUNTER_KEYS, [int(v) for v in s_vec.tolist()]))},
                          "strong_literal": literal,
                          "note": "launcher / sharding / reduction self-test on gloo; no kernel ran, no rate"}),
              flush=True)
    dist.destroy_process_group()


# ---- main -------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.profile_spec:
        specs = profile_specs()
        if args.profile_spec == "list":
            print(" ".join(specs))
            return
        sp = specs[args.profile_spec]
        print("--config %s --dtype %s --demod %s --batch %d%s" % (sp["config"], sp["dtype"], sp["demod"], sp["batch"],
                                                                 "".join(" --opt " + o for o in sp["opts"])))
        return
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    if args.launch_check:
        return launch_check(args, rank, world)
    from pyphysim_amd.engine import Engine   # loads libmcle (shares torch's HIP runtime)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    gpu = local_rank
    if local_rank >= torch.cuda.device_count():
        if not args.share_gpus:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    use_dist = world > 1 or "RANK" in os.environ        # any torch.distributed.run launch, even with 1 rank
    xdev = "cuda" if args.dist_backend == "nccl" else "cpu"      # where the exchanged tensors live
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group("gloo")
    eng = Engine(gpu, args.dtype)
    for item in args.opt:
        name, _, val = item.partition("=")
        eng.set_option(name, int(val))
        ACTIVE_OPTS[name] = int(val)
    batch = args.batch or BATCH[args.config]
    exchange_calls = {"timed": 0}
    # ---- who carries the exchange: the product's own RCCL communicator (csrc/comm.hip) or torch.distributed ----
    native, comm_note = None, None
    if use_dist and args.comm != "torch":
        if args.dist_backend != "nccl":
            comm_note = "native exchange needs one GPU per rank (RCCL): backend is %s" % args.dist_backend
            if args.comm == "native":
                raise SystemExit("bench.py: --comm native: " + comm_note)
        else:
            from pyphysim_amd.distributed import NativeComm
            uid, ok, err = None, 1, None
            try:     # step 1: the 128-byte RCCL id, rank 0 -> everyone over the product's TCP rendezvous
                uid = NativeComm.rendezvous(eng, rank, world, timeout=60.0)
            except Exception as exc:
                ok, err = 0, repr(exc)
            flag = torch.tensor([ok], dtype=torch.int64, device=xdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank must hold the id before anyone blocks in ncclCommInitRank
            if int(flag[0]) == 1:
                # step 2: mcle_comm_init (ncclCommInitRank) on the context's device, on THIS thread -- the context is never touched
                # by two threads (ADVICE r04 #4 / VERDICT r05 item 8: rounds 4-5 ran the bring-up on a side thread and fell back
                # past a stuck one).  A bring-up that RETURNS an error falls back to torch.distributed on every rank (agreed by the
                # MIN all-reduce below); one that HANGS is ended loudly by the watchdog: exit code 3 and a line on stderr, so the
                # launcher tears the job down instead of timing a run whose exchange never came up.
                import threading
                done = threading.Event()

                def _watchdog():
                    if not done.wait(float(os.environ.get("MCLE_BENCH_COMM_DEADLINE_S", "120"))):
                        sys.stderr.write("bench.py: rank %d: mcle_comm_init (ncclCommInitRank) did not return within its "
                                         "deadline; aborting the run (exit 3)\n" % rank)
                        sys.stderr.flush()
                        os._exit(3)
                threading.Thread(target=_watchdog, daemon=True).start()
                comm, cerr = None, None
                try:
                    comm = NativeComm(eng, rank, world, unique_id=uid)
                except Exception as exc:            # noqa: BLE001 -- reported in the line
                    cerr = repr(exc)
                done.set()
                flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=xdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag[0]) == 1:
                    native = comm
                else:
                    comm_note = ("mcle_comm_init failed on some rank (%s): exchange through torch.distributed"
                                 % (cerr or "this rank ok"))
                    if args.comm == "native":
                        raise SystemExit("bench.py: --comm native: " + comm_note)
                    if comm is not None:               # a rank whose communicator DID come up destroys it (ncclCommDestroy)
                        try:
                            comm.close()
                        except Exception:          # noqa: BLE001 -- the fallback must go on
                            pass
            else:
                comm_note = "NativeComm rendezvous failed on some rank (%s): exchange through torch.distributed" % err
                if args.comm == "native":
                    raise SystemExit("bench.py: --comm native: " + comm_note)
    exchange_impl = ("native: mcle_counters_allreduce (pack kernel + grouped ncclAllReduce uint64 SUM / MAX + unpack on the "
                     "context stream; the counters never visit the host before the reduction)") if native else \
                    ("torch: read_counters -> torch.distributed.all_reduce(%s)" % args.dist_backend if use_dist else None)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def timed(demod, dtype, base, solo=False):
        """W warm-up + exactly K timed steps of the hot path on this rank's contiguous index range starting at
        `base`, one all-reduce of the counter vector inside the timed region -> dict(elapsed max over ranks, kernel ms
        max / min over ranks, reduced counter totals, workload description, units per realization).
        solo: only rank 0 launches (the other ranks hold the barriers and contribute zeros): the one-GPU rate measured inside
        the same multi-rank job, so that N-GPU efficiency can be read off one line."""
        run, units, workload = make_runner(eng, args.config, demod, dtype)
        active = (not solo) or rank == 0
        counters = eng.new_counters()
        n_active = 1 if solo else world
        r_idx = 0 if solo else rank
        if active and args.preroll_ms > 0:    # clock ramp: untimed, results discarded, indices far from everything else
            t_pre, i_pre = time.perf_counter(), 0
            while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
                run((1 << 41) + base + (i_pre * n_active + r_idx) * batch, batch, counters)
                eng.sync()
                i_pre += 1
        if active:
            for w in range(args.warmup):      # warm-up draws from a disjoint index range far away
                run((1 << 40) + base + (w * n_active + r_idx) * batch, batch, counters)
        if use_dist:   # bring the communicator up outside the timed region (same shape / dtype as the real exchange)
            dist.all_reduce(torch.zeros(6, dtype=torch.int64, device=xdev), op=dist.ReduceOp.SUM)
            dist.all_reduce(torch.zeros(2, dtype=torch.float64, device=xdev), op=dist.ReduceOp.MAX)
            if native is not None:
                eng.counters_allreduce(counters, 1)
        barrier()
        counters.zero()
        barrier()
        lo = base + r_idx * args.steps * batch
        t0 = time.perf_counter()
        kernel_ms = 0.0
        if active:
            eng.timer_start()
            for s in range(args.steps):
                run(lo + s * batch, batch, counters)
            kernel_ms = eng.timer_stop_ms()        # HIP events on the stream the kernels ran on
        if native is not None:
            eng.counters_allreduce(counters, 1)            # RCCL over xGMI, on the context stream, in place: the path's only
            exchange_calls["timed"] += 1                   # exchange step, as BatchedSimulationRunner + NativeComm run it
            local = eng.read_counters(counters)            # (blocks on the stream) -- every rank now reads the global sums
            vec = torch.tensor([local[k] for k in COUNTER_KEYS], dtype=torch.int64, device=xdev)
        else:
            local = eng.read_counters(counters)
            vec = torch.tensor([local[k] for k in COUNTER_KEYS], dtype=torch.int64, device=xdev)
            if use_dist:
                dist.all_reduce(vec, op=dist.ReduceOp.SUM)     # RCCL over xGMI: the path's only exchange step
                exchange_calls["timed"] += 1
        barrier()
        elapsed = time.perf_counter() - t0
        tmax = torch.tensor([elapsed, kernel_ms, -kernel_ms if active else -1e30], dtype=torch.float64, device=xdev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = [int(v) for v in vec.tolist()]
        assert tot[0] + tot[1] == args.steps * batch * n_active, (tot, args.steps, batch, n_active)
        return {"elapsed": float(tmax[0]), "kernel_ms": float(tmax[1]), "kernel_ms_min": -float(tmax[2]), "tot": tot,
                "workload": workload, "units": units, "rate": (tot[0] + tot[1]) / float(tmax[0]),
                "kernel_ms_per_launch": float(tmax[1]) / args.steps}

    def timed_strong(demod, dtype, total, base, solo=False, cfg=None):
        """The configuration's LITERAL realization count split over the ranks ("strong" scaling): rank r runs the contiguous
        block [floor(total r / n), floor(total (r + 1) / n)) in calls of <= batch realizations, then the one all-reduce;
        barrier + synchronize on both sides; --strong-reps repetitions on disjoint index ranges after one untimed one,
        each repetition's elapsed time = max over ranks.  cfg: another configuration than --config (the line's strong_c5 block)."""
        run, units, workload = make_runner(eng, cfg or args.config, demod, dtype)
        s_batch = batch if cfg is None else BATCH[cfg]
        n_active = 1 if solo else world
        r_idx = 0 if solo else rank
        active = (not solo) or rank == 0
        counters = eng.new_counters()
        times = []
        for rep in range(args.strong_reps + 1):
            lo = base + rep * total + (total * r_idx) // n_active
            hi = base + rep * total + (total * (r_idx + 1)) // n_active
            barrier()
            counters.zero()
            barrier()
            t0 = time.perf_counter()
            if active:
                pos = lo
                while pos < hi:
                    n = min(s_batch, hi - pos)
                    run(pos, n, counters)
                    pos += n
            if native is not None:
                eng.counters_allreduce(counters, 1)
                local = eng.read_counters(counters)
                vec = torch.tensor([local[k] for k in COUNTER_KEYS], dtype=torch.int64, device=xdev)
            else:
                local = eng.read_counters(counters)
                vec = torch.tensor([local[k] for k in COUNTER_KEYS], dtype=torch.int64, device=xdev)
                if use_dist:
                    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
            barrier()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=xdev)
            if use_dist:
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
            tot_s = [int(v) for v in vec.tolist()]
            assert tot_s[0] + tot_s[1] == total, (tot_s, total)
            if rep > 0:
                times.append(float(el[0]))
        times.sort()
        return {"elapsed_s": times[len(times) // 2], "elapsed_min_s": times[0], "elapsed_all_s": times,
                "ser": tot_s[2] / float(max(1, tot_s[0]) * units)}

    head = timed(args.demod, args.dtype, 0)
    elapsed, kernel_ms, tot, workload, units = head["elapsed"], head["kernel_ms"], head["tot"], head["workload"], head["units"]
    n_real = tot[0] + tot[1]
    # the same workload in the other arithmetic and with the other demodulator, each timed exactly like the headline
    rates = {args.dtype: {args.demod: head}}
    if not args.single_demod:
        base_i = 1
        for dt in (args.dtype, "f32" if args.dtype == "f64" else "f64"):
            for dm in ((args.demod, "slicer" if args.demod == "mindist" else "mindist") if args.config == "c4" else (args.demod,)):
                if dt == args.dtype and dm == args.demod:
                    continue
                rates.setdefault(dt, {})[dm] = timed(dm, dt, base_i << 36)
                base_i += 1
    solo = timed(args.demod, args.dtype, 15 << 36, solo=True) if world > 1 else None
    strong = None
    if use_dist and args.strong_reps > 0:
        s_total = args.strong_total or STRONG_TOTAL[args.config]
        s_all = timed_strong(args.demod, args.dtype, s_total, 17 << 36)
        s_one = timed_strong(args.demod, args.dtype, s_total, 18 << 36, solo=True) if world > 1 else s_all
        strong = {"scaling": "strong", "total_realizations": s_total,
                  "per_rank_realizations": [(s_total * (r + 1)) // world - (s_total * r) // world for r in range(world)],
                  "elapsed_s": s_all["elapsed_s"], "elapsed_min_s": s_all["elapsed_min_s"], "reps": args.strong_reps,
                  "value": s_total / s_all["elapsed_s"], "unit": "realizations/s",
                  "n1_elapsed_s": s_one["elapsed_s"], "n1_value": s_total / s_one["elapsed_s"],
                  "speedup_vs_n1": s_one["elapsed_s"] / s_all["elapsed_s"],
                  "efficiency": s_one["elapsed_s"] / s_all["elapsed_s"] / world,
                  "ser": s_all["ser"],
                  "what": "BASELINE.json's literal realization count for this configuration split contiguously over the ranks, "
                          "one counter all-reduce, barrier + synchronize on both sides, median of the repetitions; n1 = rank 0 "
                          "alone on the same total inside the same job",
                  "note": "at this size a rank's share is %.2f ms of kernel at the one-rank rate: launch latency, the all-reduce and "
                          "the two barriers are a visible part of the region, so the strong figure sits below the weak one by "
                          "construction" % (1e3 * s_one["elapsed_s"] / world)}
    # BASELINE config 5 ("1e5 realizations on 8 MI355X") gets its strong block on the config-4 line as well: the two configurations
    # BASELINE.json shards over the node are both on the one line the driver collects per N
    strong_c5 = None
    if strong is not None and args.config == "c4":
        t5 = STRONG_TOTAL["c5"]
        a5 = timed_strong(args.demod, args.dtype, t5, 21 << 36, cfg="c5")
        o5 = timed_strong(args.demod, args.dtype, t5, 22 << 36, solo=True, cfg="c5") if world > 1 else a5
        strong_c5 = {"scaling": "strong", "config": "c5", "total_realizations": t5,
                     "per_rank_realizations": [(t5 * (r + 1)) // world - (t5 * r) // world for r in range(world)],
                     "elapsed_s": a5["elapsed_s"], "elapsed_min_s": a5["elapsed_min_s"], "reps": args.strong_reps,
                     "value": t5 / a5["elapsed_s"], "unit": "realizations/s", "n1_elapsed_s": o5["elapsed_s"],
                     "n1_value": t5 / o5["elapsed_s"], "speedup_vs_n1": o5["elapsed_s"] / a5["elapsed_s"],
                     "efficiency": o5["elapsed_s"] / a5["elapsed_s"] / world, "ser": a5["ser"],
                     "what": "BASELINE config 5 (K=3 2x2 closed-form IA + 16-QAM, 200 symbols per stream, 20 dB): its literal 10^5 "
                             "realizations split contiguously over the ranks, one counter all-reduce, median of the repetitions",
                     "note": "10^5 realizations are %.2f ms of kernel at the one-rank rate: the region is launch latency, the "
                             "all-reduce and two barriers -- a latency figure, not a throughput one" % (1e3 * o5["elapsed_s"])}
        make_runner(eng, args.config, args.demod, args.dtype)      # back to this line's constellation
    # who ran: device name and PCI bus id of every rank (RCCL's view of the job next to the launcher's)
    props = torch.cuda.get_device_properties(gpu)
    me = {"rank": rank, "local_rank": local_rank, "gpu": gpu, "device": props.name,
          "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                              getattr(props, "pci_device_id", 0)),
          "kernel_ms_timed_region": head["kernel_ms"] if world == 1 else None}
    ranks_info = [me]
    if use_dist:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)

    if rank == 0:
        value = n_real / elapsed
        if world == 1 and not args.no_cpu:
            try:
                sys.path.insert(0, os.path.join(REPO, "scripts"))
                import bench_staged_c4
                MEASURED["stream"] = bench_staged_c4.measure_hbm_stream(eng)
                MEASURED["copy_GBps"] = MEASURED["stream"]["achievable_GBps"]
            except Exception:
                pass

        def roof(dt, dm, res):
            """roofline block of the (dtype, demod) kernel: live rocprofv3 counters when possible (child runs at a smaller
            batch: the counters are normalised per realization), else the committed summary"""
            per_launch_s = res["kernel_ms_per_launch"] * 1e-3
            pmc, src, pmc_batch = None, None, None
            want = args.pmc == "on" or (args.pmc == "auto" and world == 1 and not args.no_cpu)
            if want:
                eng.sync()
                pmc_batch = min(batch, BATCH_SURVEY[args.config] * 4)
                pmc, err = collect_pmc_live(args, pmc_batch, dt, dm)
                src = ("rocprofv3 --pmc child runs of this command (%d steps of %d realizations per pass, every dispatch of the kernel "
                       "summed: %s)" % (PMC_CHILD_STEPS, pmc_batch, "; ".join(n for _, n in PMC_PASSES))) if pmc else \
                    "live collection failed (%s); " % err
            if pmc is None:
                pmc, path, pmc_batch = committed_pmc(args.config, dt)
                src = ((src or "") + "%s (committed rocprofv3 summary, NOT measured in this run)" % path) if pmc else src
            live = src is not None and src.startswith("rocprofv3 --pmc child")
            d = derive_pmc({k: v for k, v in pmc.items() if not k.startswith("_")},
                           pmc_batch * (PMC_CHILD_STEPS if live else 1)) if pmc else {}
            if live and pmc:
                d["dispatches_per_step"] = pmc.get("_dispatches_sq", 0) / float(PMC_CHILD_STEPS)
            blk = roofline_block(args, dt, batch, per_launch_s, batch / per_launch_s, d, src)
            blk["demod"] = dm
            blk["counters_realizations_per_launch"] = pmc_batch
            return blk

        dtype_rates = {dt: r[args.demod]["rate"] for dt, r in rates.items() if args.demod in r}
        demod_rates = {dm: r["rate"] for dm, r in rates[args.dtype].items()}
        out = {
            "metric": "Monte Carlo realizations/sec (whole node) + SER abs-error vs ref",
            "value": value, "unit": "realizations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preroll_ms": args.preroll_ms,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "value_is": "%s arithmetic (%s), %s demodulator" % (
                args.dtype, "complex128 = the reference's own precision" if args.dtype == "f64" else "complex64",
                "min-distance search over the LDS constellation table (north star)" if args.demod == "mindist" else "QAM slicer"),
            "timed_region_s": elapsed,
            "dtype_rates": dtype_rates,
            "rates": {dt: {dm: r["rate"] for dm, r in v.items()} for dt, v in rates.items()},
            "kernel_ms_per_launch": {dt: {dm: r["kernel_ms_per_launch"] for dm, r in v.items()} for dt, v in rates.items()},
            "config": {"workload": workload, "realizations_per_step_per_gpu": batch,
                       "demod": args.demod, "demod_rates": demod_rates,
                       "demod_note": "mindist = min-distance search over the LDS constellation table (north star); "
                                     "slicer = QAM slicer, decision-identical in f64",
                       "symbols_per_realization": units, "parallelism": "realization-sharded x%d" % world,
                       "rank_ranges": [[r * args.steps * batch, (r + 1) * args.steps * batch] for r in range(world)],
                       "exchange": ("one all-reduce(SUM) of 6 int64 counters over RCCL, inside the timed region "
                                    "(%d ranks; %s)" % (world, "the product's mcle_counters_allreduce" if native is not None
                                                        else "torch.distributed.all_reduce"))
                       if use_dist else "none (single process, no process group)",
                       "rng": "Philox4x32-10 keyed by (seed, realization)"},
            "rccl": {"process_group": bool(use_dist), "backend": dist.get_backend() if use_dist else None,
                     "exchange_impl": exchange_impl, "exchange_note": comm_note,
                     "native_comm": ({"rank": native.rank, "world": native.world, "info": list(eng.comm_info())}
                                     if native is not None else None),
                     "rccl_world_size": dist.get_world_size() if use_dist else 1,
                     "allreduce_calls_in_timed_regions": exchange_calls["timed"],
                     "ranks": ranks_info},
            "ser": tot[2] / float(max(1, tot[0]) * units),
            "ber": tot[4] / float(max(1, tot[0]) * units * BITS[args.config]),
            "n_skipped": tot[1],
            "roofline": roof(args.dtype, args.demod, head),
        }
        if strong is not None:
            out["strong"] = strong
        if strong_c5 is not None:
            out["strong_c5"] = strong_c5
        if world > 1:
            out["kernel_ms_per_rank"] = {"min": head["kernel_ms_min"], "max": head["kernel_ms"]}
            out["n1_value"] = solo["rate"]
            out["n1_note"] = ("rank 0 alone, same job, same K steps and batch, the other ranks holding the barriers; "
                              "value / (n_gpus * n1_value) = %.4f" % (value / (world * solo["rate"])))
        for dt, v in rates.items():
            if dt != args.dtype and args.demod in v:
                out["roofline_" + dt] = roof(dt, args.demod, v[args.demod])
        out["roofline_" + args.dtype] = out["roofline"]
        for dm, r in rates[args.dtype].items():
            if dm != args.demod:
                out[dm + "_demod_realizations_per_s"] = r["rate"]
                out[dm + "_demod_kernel_ms_per_launch"] = r["kernel_ms_per_launch"]
        if world == 1 and not args.no_cpu:
            # per-realization counts of the first realizations for the SER cross-check
            res, se, be = eng_first_counts(eng, args, 16384 if args.config != "c2" else 128)
            cb, ser_err, n_chk = cpu_baseline(args.config, args.cpu_seconds, se)
            cb["host_cpu_count"] = os.cpu_count()
            cb["host_cpu_model"] = _cpu_model()
            cb["dtype"] = "complex128 (NumPy)"
            cross = os.path.join(REPO, "profiles", "cpu_cross_timing.json")
            if os.path.exists(cross):
                try:
                    cb["reference_vs_port"] = json.load(open(cross)).get("summary")
                except Exception:
                    pass
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
            if args.config == "c4":
                # the other workloads of SURVEY.md section 8 on the same device, 5 launches each (a second's work):
                # realizations/s and the kernel time of one launch in both arithmetics, so that one bench line documents them all
                others = {}
                for cfg in ("c2", "c3", "c5", "f1", "f6"):
                    others[cfg] = {}
                    for dt in ("f64", "f32"):
                        try:
                            run_o, units_o, wl_o = make_runner(eng, cfg, "slicer", dt)
                            cnt_o = eng.new_counters()
                            nb = BATCH_SURVEY[cfg]
                            # untimed launches until the device has been busy for ~60 ms (the legs are 1 - 50 ms of kernels each
                            # and follow seconds of host-only work: timed cold, a leg of five 0.9 ms launches read 5 - 8 % low
                            # and moved by that much from run to run)
                            t_warm, n_warm = time.time(), 0
                            while n_warm < 64 and (n_warm < 1 or time.time() - t_warm < 0.06):
                                run_o((1 << 42) - (n_warm + 1) * nb, nb, cnt_o)
                                eng.sync()
                                n_warm += 1
                            cnt_o = eng.new_counters()
                            eng.timer_start()
                            for s2 in range(5):
                                run_o((1 << 42) + (s2 + 1) * nb, nb, cnt_o)
                            ms_o = eng.timer_stop_ms() / 5
                            c_o = eng.read_counters(cnt_o)
                            rate_o = nb / ms_o * 1e3
                            others[cfg][dt] = {"realizations_per_s": rate_o, "kernel_ms_per_launch": ms_o,
                                               "realizations_per_launch": nb, "kernel": kernel_name(cfg, dt),
                                               "flop_frac": sum(flops_for(cfg, dt).values()) * rate_o / 1e12 / PEAK_TFLOPS[dt],
                                               "flop_frac_survey_model": (sum(FLOPS[cfg].values()) * rate_o / 1e12 / PEAK_TFLOPS[dt])
                                               if cfg in ("c3", "f1") else None,
                                               "flops_per_realization": sum(flops_for(cfg, dt).values()),
                                               "ser": c_o["sym_errors"] / float(max(1, c_o["n_realizations"]) * units_o)}
                            others[cfg]["workload"] = wl_o
                        except Exception as exc:
                            others[cfg][dt] = {"error": repr(exc)}
                # the north star's matrix-core clause ("MFMA only for the batched Nt x Nr x Ns MIMO contraction ... evidenced by
                # MFMA-busy"): config 4 in complex64 on the matrix-core kernel (option f32_mfma = 1: H.X / G.Y as
                # v_mfma_f32_4x4x1, the DFT-16 passes as v_mfma_f32_16x16x4; the planar VALU kernel above is 10-20 % faster and the
                # default), with its live MFMA-busy and VALU-busy counters, measured in THIS run
                try:
                    import copy as _copy
                    with eng.options(f32_mfma=1):
                        run_m, units_m, wl_m = make_runner(eng, "c4", args.demod, "f32")
                        cnt_m = eng.new_counters()
                        nb = BATCH_SURVEY["c4"]
                        run_m(1 << 45, nb, cnt_m)
                        eng.sync()
                        eng.timer_start()
                        for s2 in range(5):
                            run_m((1 << 45) + (s2 + 1) * nb, nb, cnt_m)
                        ms_m = eng.timer_stop_ms() / 5
                        c_m = eng.read_counters(cnt_m)
                    rate_m = nb / ms_m * 1e3
                    leg = {"realizations_per_s": rate_m, "kernel_ms_per_launch": ms_m, "realizations_per_launch": nb,
                           "kernel": "k_run_mimo_ofdm_mfma", "option": "f32_mfma=1", "demod": args.demod, "workload": wl_m,
                           "flop_frac": sum(flops_for("c4", "f32").values()) * rate_m / 1e12 / PEAK_TFLOPS["f32"],
                           "ser": c_m["sym_errors"] / float(max(1, c_m["n_realizations"]) * units_m)}
                    if args.pmc != "off":
                        eng.sync()
                        args_m = _copy.copy(args)
                        args_m.opt = list(args.opt) + ["f32_mfma=1"]
                        ACTIVE_OPTS["f32_mfma"] = 1
                        try:
                            pmc_m, err_m = collect_pmc_live(args_m, nb, "f32", args.demod)
                        finally:
                            ACTIVE_OPTS.pop("f32_mfma", None)
                        if pmc_m:
                            dm_ = derive_pmc({k: v for k, v in pmc_m.items() if not k.startswith("_")}, nb * PMC_CHILD_STEPS)
                            leg.update(mfma_busy_chip=dm_.get("mfma_busy_chip"),
                                       valu_busy_chip=dm_.get("valu_issue_frac") if dm_.get("valu_issue_frac") is not None
                                       else (min(1.0, dm_["valu_busy_chip"]) if dm_.get("valu_busy_chip") is not None else None),
                                       valu_active_x4_over_simd_cycles=dm_.get("valu_busy_chip"),
                                       mfma_f32_mops_per_realization=dm_.get("mfma_f32_mops_per_realization"),
                                       valu_wave_insts_per_realization=dm_.get("valu_wave_insts_per_realization"),
                                       counters_source="rocprofv3 --pmc child runs of this command with --opt f32_mfma=1")
                        else:
                            leg["counters_source"] = "live collection failed (%s)" % err_m
                    others["c4_f32_mfma"] = leg
                except Exception as exc:
                    others["c4_f32_mfma"] = {"error": repr(exc)}
                # the north star's HBM clause: config 4 staged through HBM, both arithmetics, with the copy rate of THIS box
                # measured in THIS run and the bytes the chain really moves (rocprofv3 child runs; SURVEY 8(d)'s min rule)
                try:
                    sys.path.insert(0, os.path.join(REPO, "scripts"))
                    import bench_staged_c4
                    copy_bw = MEASURED["copy_GBps"] or bench_staged_c4.measure_copy_GBps(eng)
                    want_bytes = args.pmc != "off"
                    others["c4_staged"] = bench_staged_c4.run(eng, batch=8192, seconds=1.0, copy_GBps=copy_bw,
                                                              hbm_counters=want_bytes, stream=MEASURED["stream"])
                    others["c4_staged_f64"] = bench_staged_c4.run(eng, batch=8192, seconds=1.0, dtype="f64", copy_GBps=copy_bw,
                                                                  hbm_counters=want_bytes, stream=MEASURED["stream"])
                    out["hbm_copy_GBps_measured_this_run"] = copy_bw
                    out["hbm_stream_rates_measured_this_run"] = MEASURED["stream"]
                except Exception as exc:
                    others["c4_staged"] = {"error": repr(exc)}
                out["other_workloads"] = others
        print(json.dumps(out), flush=True)
    if native is not None:
        native.close()
    if use_dist:
        dist.destroy_process_group()
    eng.close()


def eng_first_counts(eng, args, n):
    from pyphysim_amd import _lib
    from pyphysim_amd.modulators import constellation
    method = _lib.DEMOD_QAM_SLICER if args.demod == "slicer" else _lib.DEMOD_MINDIST
    nv = 1.0 / (10.0 ** (SNR_DB[args.config] / 10.0))
    make_runner(eng, args.config, args.demod, args.dtype)      # (re)binds this configuration's constellation
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
