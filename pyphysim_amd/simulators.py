"""Ready-made link simulators on the fused GPU pipelines, shaped like the reference's own template
simulators (apps/awgn_modulators/simulate_psk.py, apps/mimo/simulate_mimo.py,
notebooks/TDL_and_OFDM.ipynb) but advancing by batches: subclasses of BatchedSimulationRunner
whose ``_run_batch`` is one call into libmcle.

    sim = MimoOfdmSimulator(SNR=[15, 20, 25], rep_max=100000)
    sim.simulate()
    sim.results.get_result_values_list('ser')
"""
import numpy as np

from . import _lib
from .channels import discretize_profile
from .engine import get_engine
from .modulators import BPSK, PSK, QAM, QPSK, dB2Linear
from .simulations import BatchedSimulationRunner

_GOLDEN = 0x9E3779B97F4A7C15


def make_modulator(name, M=None, engine=None):
    name = name.lower()
    if name == "qam":
        return QAM(M, engine=engine)
    if name == "psk":
        return PSK(M, engine=engine)
    if name == "qpsk":
        return QPSK(engine=engine)
    if name == "bpsk":
        return BPSK(engine=engine)
    raise ValueError("unknown modulator %r" % (name,))


class _LinkSimulator(BatchedSimulationRunner):
    """Common part: SNR sweep ('SNR' unpacked), modulator, seed handling, engine binding."""

    def __init__(self, SNR, modulator="qam", M=16, rep_max=1000, seed=0, batch_size=4096, dtype="f32",
                 demod="auto", engine=None, common_random_numbers=False, process_group=None,
                 exact_early_stop=False):
        super().__init__(batch_size=batch_size, process_group=process_group, exact_early_stop=exact_early_stop)
        self.rep_max = rep_max
        self.seed = int(seed)
        self.dtype = dtype
        self.common_random_numbers = common_random_numbers
        self._engine = engine
        self.modulator = make_modulator(modulator, M, engine) if isinstance(modulator, str) else modulator
        if demod == "auto":
            demod = "slicer" if isinstance(self.modulator, QAM) else "mindist"
        self.demod_method = _lib.DEMOD_QAM_SLICER if demod == "slicer" else _lib.DEMOD_MINDIST
        self.params.add("SNR", np.atleast_1d(np.asarray(SNR, dtype=float)))
        self.params.set_unpack_parameter("SNR")
        self.params.add("modulator", self.modulator.name)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def _seed_for(self, current_parameters):
        """Independent randomness per parameter variation (the reference keeps drawing from one
        global stream) unless common_random_numbers asks for the same draws at every SNR."""
        if self.common_random_numbers:
            return self.seed
        return (self.seed + (max(current_parameters.unpack_index, 0) + 1) * _GOLDEN) & 0xFFFFFFFFFFFFFFFF

    def _bind(self):
        self.modulator._engine = self.engine
        return self.modulator._bind()

    @staticmethod
    def _noise_var(current_parameters):
        return 1.0 / float(dB2Linear(current_parameters["SNR"]))

    def _launch(self, current_parameters, first_rep, count, per_realization):
        raise NotImplementedError

    def _run_batch(self, current_parameters, first_rep, count):
        return self._launch(current_parameters, first_rep, count, False)

    def _run_batch_detailed(self, current_parameters, first_rep, count):
        return self._launch(current_parameters, first_rep, count, True)[:3]


class AwgnSimulator(_LinkSimulator):
    """Config 1: apps/awgn_modulators/simulate_psk.py:51-115."""

    def __init__(self, SNR, modulator="qam", M=16, NSymbs=10000, **kw):
        super().__init__(SNR, modulator, M, **kw)
        self.params.add("NSymbs", int(NSymbs))

    def _launch(self, current_parameters, first_rep, count, per_realization):
        eng = self._bind()
        return eng.run_awgn(current_parameters["NSymbs"], self._noise_var(current_parameters),
                            self._seed_for(current_parameters), first_rep, count, method=self.demod_method,
                            dtype=self.dtype, per_realization=per_realization)


class FlatFadingSimulator(_LinkSimulator):
    """Config 2: flat fading SuChannel(JakesSampleGenerator(Fd, Ts, L)) (or i.i.d. Rayleigh),
    receiver equalises with the known channel."""

    def __init__(self, SNR, modulator="qam", M=64, NSymbs=100000, Fd=100.0, Ts=1e-3, L=8, rayleigh_iid=False, **kw):
        super().__init__(SNR, modulator, M, **kw)
        for k, v in (("NSymbs", int(NSymbs)), ("Fd", float(Fd)), ("Ts", float(Ts)), ("L", int(L)),
                     ("rayleigh_iid", bool(rayleigh_iid))):
            self.params.add(k, v)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        eng = self._bind()
        return eng.run_flat_fading(p["NSymbs"], self._noise_var(p), self._seed_for(p), first_rep, count, Fd=p["Fd"],
                                   Ts=p["Ts"], L=p["L"], rayleigh_iid=p["rayleigh_iid"], method=self.demod_method,
                                   dtype=self.dtype, per_realization=per_realization)


class OfdmTdlSimulator(_LinkSimulator):
    """Config 3: notebooks/TDL_and_OFDM.ipynb OfdmTdlSimulator (OFDM over a Jakes TDL channel with
    the one-tap equaliser)."""

    def __init__(self, SNR, modulator="qpsk", M=4, fft_size=1024, cp_size=16, num_used_subcarriers=None,
                 num_ofdm_symbols=1, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8,
                 tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays=None, **kw):
        super().__init__(SNR, modulator, M, **kw)
        if tap_delays is None:
            tap_delays = np.arange(len(tap_powers_dB)) * Ts
        self._tap_power, self._tap_delay = discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                                               np.asarray(tap_delays, dtype=float), Ts)
        for k, v in (("fft_size", int(fft_size)), ("cp_size", int(cp_size)),
                     ("num_used_subcarriers", int(num_used_subcarriers or fft_size)),
                     ("num_ofdm_symbols", int(num_ofdm_symbols)), ("Fd", float(Fd)), ("Ts", float(Ts)),
                     ("L", int(L))):
            self.params.add(k, v)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        eng = self._bind()
        return eng.run_ofdm_tdl(p["fft_size"], p["cp_size"], p["num_used_subcarriers"], p["num_ofdm_symbols"],
                                self._noise_var(p), self._tap_power, self._tap_delay, self._seed_for(p), first_rep,
                                count, Fd=p["Fd"], Ts=p["Ts"], L=p["L"], method=self.demod_method, dtype=self.dtype,
                                per_realization=per_realization)


class MimoOfdmSimulator(_LinkSimulator):
    """Config 4: apps/mimo/simulate_mimo.py:68-142 (Blast, flat H = randn_c(Nr, Nt) per
    realization) with per-antenna OFDM; MMSE (set_noise_var) or zero forcing."""

    def __init__(self, SNR, modulator="qam", M=64, Nt=4, Nr=4, fft_size=1024, cp_size=16,
                 num_used_subcarriers=None, num_ofdm_symbols=1, mmse=True, **kw):
        super().__init__(SNR, modulator, M, **kw)
        for k, v in (("Nt", int(Nt)), ("Nr", int(Nr)), ("fft_size", int(fft_size)), ("cp_size", int(cp_size)),
                     ("num_used_subcarriers", int(num_used_subcarriers or fft_size)),
                     ("num_ofdm_symbols", int(num_ofdm_symbols)), ("mmse", bool(mmse))):
            self.params.add(k, v)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        eng = self._bind()
        return eng.run_mimo_ofdm(p["Nt"], p["Nr"], p["fft_size"], p["cp_size"], p["num_used_subcarriers"],
                                 p["num_ofdm_symbols"], self._noise_var(p), self._seed_for(p), first_rep, count,
                                 mmse=p["mmse"], method=self.demod_method, dtype=self.dtype,
                                 per_realization=per_realization)


class MimoSimulator(_LinkSimulator):
    """apps/mimo/simulate_mimo.py:22-142 (MIMOSimulationRunner and its Alamouti / Blast / MRC / MRT / SVDMimo /
    GMDMimo subclasses): flat channel per realization, NSymbs symbols per layer, single carrier.  `mmse=True`
    is Blast.set_noise_var(noise_var) (the app itself runs zero forcing)."""

    def __init__(self, SNR, scheme="blast", modulator="qam", M=16, Nt=2, Nr=2, NSymbs=200, mmse=False, **kw):
        super().__init__(SNR, modulator, M, **kw)
        if scheme not in _lib.MIMO_SCHEMES:
            raise ValueError("unknown MIMO scheme %r" % (scheme,))
        for k, v in (("scheme", scheme), ("Nt", int(Nt)), ("Nr", int(Nr)), ("NSymbs", int(NSymbs)),
                     ("mmse", bool(mmse))):
            self.params.add(k, v)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        eng = self._bind()
        return eng.run_mimo_flat(p["scheme"], p["Nt"], p["Nr"], p["NSymbs"], self._noise_var(p), self._seed_for(p),
                                 first_rep, count, mmse=p["mmse"], method=self.demod_method, dtype=self.dtype,
                                 per_realization=per_realization)


class BdSimulator(_LinkSimulator):
    """apps/comp_BD/simulate_comp_simple.py:22-140 (no external interference source): K cells with Nr x Nr antennas
    each transmit jointly through a block-diagonalising precoder (water-filling normalised to the strongest cell,
    BlockDiagonalizer.block_diagonalize) and every user zero-forces with pinv(newH).  As in the app the noise
    power is fixed and SNR sets the transmit power: iPu = dB2Linear(SNR) * noise_var / path_loss_border."""

    def __init__(self, SNR, modulator="psk", M=4, K=3, Nr=2, NSymbs=500, noise_var=None, path_loss_border=1.0,
                 pathloss=None, bd_noise_var=1e-50, waterfilling=True, **kw):
        super().__init__(SNR, modulator, M, **kw)
        for k, v in (("K", int(K)), ("Nr", int(Nr)), ("NSymbs", int(NSymbs)), ("waterfilling", bool(waterfilling))):
            self.params.add(k, v)
        # conversion.dBm2Linear(-116.4): the app's N0 (simulate_comp_simple.py:42,62)
        self.noise_var = float(noise_var) if noise_var is not None else 10.0 ** ((-116.4 - 30.0) / 10.0)
        self.path_loss_border = float(path_loss_border)
        self.pathloss = None if pathloss is None else np.asarray(pathloss, dtype=float)
        self.bd_noise_var = float(bd_noise_var)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        eng = self._bind()
        iPu = float(dB2Linear(p["SNR"])) * self.noise_var / self.path_loss_border
        return eng.run_bd(p["K"], p["Nr"], p["NSymbs"], iPu, self.noise_var, self._seed_for(p), first_rep, count,
                          bd_noise_var=self.bd_noise_var, pathloss=self.pathloss, waterfilling=p["waterfilling"],
                          method=self.demod_method, dtype=self.dtype, per_realization=per_realization)


class MimoOfdmTdlSimulator(_LinkSimulator):
    """spatial multiplexing over a frequency-selective MIMO TDL channel
    (TdlMimoChannel fading.py:1290-1333, MIMO branch of corrupt_data :1107-1117), per-antenna OFDM and one
    MMSE/ZF receive filter per used subcarrier from the per-symbol mean frequency response (:513-536).

    Unlike configs 1-5 this chain is STAGED: a batch of realizations flows through the batched operator
    kernels with every intermediate resident in HBM (symbols -> modulate -> Blast -> OFDM -> Jakes taps ->
    TDL -> AWGN -> OFDM^-1 -> mean response -> filters -> decode -> demod+count); nothing but the counters
    returns to the host.  Draw layout: the mcle-philox-v1 streams of realization r (DATA symbols, PHASE
    (L, S, Nr, Nt) phi then psi, NOISE [Nr, n + max delay])."""

    def __init__(self, SNR, modulator="qam", M=16, Nt=2, Nr=2, fft_size=64, cp_size=16, num_used_subcarriers=None,
                 num_ofdm_symbols=2, Fd=50.0, Ts=1e-6, L=8, tap_powers_dB=(0.0, -4.0, -9.0), tap_delays=None,
                 mmse=True, fused="auto", **kw):
        self.fused = fused                    # "auto": fused kernel when it supports the configuration
        kw.setdefault("batch_size", 4096 if fused else 256)
        super().__init__(SNR, modulator, M, **kw)
        if tap_delays is None:
            tap_delays = np.asarray((0, 2, 5)[:len(tap_powers_dB)], dtype=float) * Ts
        self._tap_power, self._tap_delay = discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                                               np.asarray(tap_delays, dtype=float), Ts)
        for k, v in (("Nt", int(Nt)), ("Nr", int(Nr)), ("fft_size", int(fft_size)), ("cp_size", int(cp_size)),
                     ("num_used_subcarriers", int(num_used_subcarriers or fft_size)),
                     ("num_ofdm_symbols", int(num_ofdm_symbols)), ("Fd", float(Fd)), ("Ts", float(Ts)),
                     ("L", int(L)), ("mmse", bool(mmse))):
            self.params.add(k, v)

    _FUSED_FFT = (64, 128, 256, 512, 1024, 2048)

    def _launch(self, current_parameters, first_rep, count, per_realization):
        p = current_parameters
        # fused kernels: every 1 <= Nt <= Nr <= 4 at fft_size 256 .. 2048 with delays <= min(256, fft_size / 2) (one receive antenna per
        # wavefront, round 5; delays beyond the prefix since round 6), Nt = Nr in {2, 4} at 64 .. 2048 otherwise; the C ABI reports anything else as unsupported
        if self.fused and 1 <= p["Nt"] <= p["Nr"] <= 4 and p["fft_size"] in self._FUSED_FFT:
            try:
                eng = self._bind()
                return eng.run_mimo_ofdm_tdl(
                    p["Nt"], p["Nr"], p["fft_size"], p["cp_size"], p["num_used_subcarriers"], p["num_ofdm_symbols"],
                    self._noise_var(p), self._tap_power, self._tap_delay, self._seed_for(p), first_rep, count,
                    Fd=p["Fd"], Ts=p["Ts"], L=p["L"], mmse=p["mmse"], method=self.demod_method, dtype=self.dtype,
                    per_realization=per_realization)
            except _lib.McleUnsupported:
                if self.fused is True:
                    raise
        elif self.fused is True:
            raise ValueError("the fused pipeline supports 1 <= Nt <= Nr <= 4 and fft_size in {64, 128, ..., 2048}")
        return self._launch_staged(p, first_rep, count, per_realization)

    def _launch_staged(self, p, first_rep, count, per_realization):
        eng = self._bind()
        dt = self.dtype
        nt, nr, fft, cp, used = p["Nt"], p["Nr"], p["fft_size"], p["cp_size"], p["num_used_subcarriers"]
        n_sym, L, Ts = p["num_ofdm_symbols"], p["L"], p["Ts"]
        noise_var, seed = self._noise_var(p), self._seed_for(p)
        ns = used * n_sym                      # data symbols per antenna
        n = n_sym * (fft + cp)                 # time samples per antenna
        delays = np.asarray(self._tap_delay, dtype=np.int32)
        S = delays.size
        if count == 0:
            c = eng.read_counters(eng.new_counters())
            return (c, None, None) if per_realization else c
        idx = eng.rand_symbols_batch(nt * ns, self.modulator.M, seed, first_rep, count)
        X = eng.blast_encode(eng.modulate(idx, dtype=dt), nt, batch=count, dtype=dt)           # [count, nt, ns]
        T = eng.ofdm_modulate(X, fft, cp, used, batch=count * nt, dtype=dt).reshape(count, nt, n)
        # Jakes time axis of a fresh generator (fading_generators.py:459-467): t_k = Ts + k*delta
        step = Ts * 1.0000000001
        delta = float(np.float64(Ts + step) - np.float64(Ts))
        amp = np.repeat(np.sqrt(np.asarray(self._tap_power, dtype=float)), nr * nt) * np.sqrt(1.0 / L)
        taps = eng.jakes_taps_philox(seed, first_rep, count, L, p["Fd"], Ts, delta, amp, n, dtype=dt)
        taps5 = taps.reshape(count, S, nr, nt, n)
        faded = eng.tdl_apply_mimo(T, taps5, delays, dtype=dt)                                  # [count, nr, n+dmax]
        R = eng.awgn_philox(faded, seed, first_rep, count, noise_var, dtype=dt)
        Rn = eng.slice_rows(R, n) if R.shape[-1] != n else R
        Y = eng.ofdm_demodulate(Rn, fft, cp, used, batch=count * nr, dtype=dt).reshape(count, nr, ns)
        Hm = eng.tdl_mean_freq_response(taps5, delays, n_sym, fft, cp, used, dtype=dt, batch=count)
        # MMSE Gram matrices are positive definite: the singular-matrix flags are only read back for ZF
        G, skipped = eng.blast_filter(Hm.reshape(count * ns, nr, nt), noise_var if p["mmse"] else 0.0, dtype=dt,
                                      read_skipped=not p["mmse"])
        est = eng.blast_decode_per_subcarrier(G.reshape(count, ns, nt, nr), Y, dtype=dt)        # [count, ns*nt]
        c, se, be = eng.demod_count(est, idx, n_real=count, method=self.demod_method, dtype=dt)
        if not p["mmse"]:
            c["n_singular_subcarriers"] = int(np.count_nonzero(skipped))
        return (c, se, be) if per_realization else c


class IaSimulator(_LinkSimulator):
    """Config 5: apps/ia/simulate_ia.py:94-245 with ClosedFormIASolver(use_best_init=True) on a
    3-user 2x2 interference channel, one stream per user.  Adds the 'sum_capacity' RATIO(x, 1)
    Result of the reference app next to the error-rate Results."""

    def __init__(self, SNR, modulator="qam", M=16, NSymbs=200, solver="closed_form", max_iterations=60,
                 relative_factor=1e-6, initialize_with="random", **kw):
        """solver: 'closed_form' (ClosedFormIASolver, use_best_init) or 'alt_min' / 'min_leakage' / 'max_sinr'
        (AlternatingMinIASolver / MinLeakageIASolver / MaxSinrIASolver with initialize_with='random';
        max_iterations defaults to the app's 60, apps/ia/simulate_ia.py:330)."""
        super().__init__(SNR, modulator, M, **kw)
        if solver not in _lib.IA_SOLVERS:
            raise ValueError("unknown IA solver %r" % (solver,))
        for k, v in (("NSymbs", int(NSymbs)), ("K", 3), ("Nr", 2), ("Nt", 2), ("Ns", 1), ("solver", solver)):
            self.params.add(k, v)
        if solver != "closed_form":
            if initialize_with not in ("random", "closed_form", "alt_min", "svd"):
                raise ValueError("initialize_with must be 'random', 'closed_form', 'alt_min' or 'svd'")
            self.params.add("max_iterations", int(max_iterations))
            self.params.add("initialize_with", initialize_with)
        self.relative_factor = float(relative_factor)
        if self.exact_early_stop:
            raise ValueError("IaSimulator carries per-batch capacity / iteration sums and does not replay single "
                             "realizations: exact_early_stop is not supported")

    # per-batch sums carried next to the integer counters: added over batches, all-reduced over ranks and kept in
    # the partial-results state, so 'sum_capacity' / 'ia_runned_iterations' are right under sharding and resume
    EXTRA_KEYS = ("sum_capacity", "sum_capacity_sq", "ia_runned_iterations", "ia_runned_iterations_sq")
    EXTRA_INT_KEYS = ("ia_runned_iterations", "ia_runned_iterations_sq")

    def _run_batch(self, current_parameters, first_rep, count):
        eng = self._bind()
        p = current_parameters
        return eng.run_ia(p["NSymbs"], self._noise_var(p), self._seed_for(p), first_rep, count,
                          method=self.demod_method, dtype=self.dtype, solver=p["solver"],
                          max_iterations=p["max_iterations"] if p["solver"] != "closed_form" else 1,
                          relative_factor=self.relative_factor,
                          initialize_with=p["initialize_with"] if p["solver"] != "closed_form" else "random")

    def _results_from_counters(self, current_parameters, c):
        from .simulations import Result
        res = super()._results_from_counters(current_parameters, c)
        n = max(int(c["n_realizations"]), 1)
        cap, cap_sq = float(c.get("sum_capacity", 0.0)), float(c.get("sum_capacity_sq", 0.0))
        res.add_result(Result.from_batch("sum_capacity", Result.RATIOTYPE, cap, n, cap, cap_sq, n))
        if current_parameters["solver"] != "closed_form":
            its, its_sq = int(c.get("ia_runned_iterations", 0)), int(c.get("ia_runned_iterations_sq", 0))
            res.add_result(Result.from_batch("ia_runned_iterations", Result.RATIOTYPE, its, n, its, its_sq, n))
        return res
