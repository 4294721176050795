"""Oracle: the per-realization chains of BASELINE.json's configs.  TEST INFRASTRUCTURE.

Each ``chain_*`` composes the oracle operators in the order (and with the RNG draw
order) of the reference's own template simulators:
  C1  apps/awgn_modulators/simulate_psk.py:51-115
  C2  notebooks/Transmission_with_Rayleigh_and_AWGN_channels.ipynb cell 8 +
      pyphysim/channels/singleuser.py:48-81,130-151
  C3  notebooks/TDL_and_OFDM.ipynb (OfdmTdlSimulator._run_simulation) /
      apps/ofdm/ofdm_tdlchannel.py:30-80
  C4  apps/mimo/simulate_mimo.py:68-142 (+ per-antenna OFDM)
and returns every intermediate so that HIP kernels can be fed injected inputs.

Randomness is pluggable:
  LegacyRng(seed)          NumPy legacy MT19937, i.e. what ``np.random.seed(seed)``
                           gives the reference (container parity / golden vectors)
  PhiloxRng(seed, r)       the mcle-philox-v1 contract (oracle/philox.py), i.e. what
                           the HIP pipelines draw on-chip (common random numbers)
"""
import math

import numpy as np

from . import channels as och
from . import ia as oia
from . import bd as obd
from . import mimo as omimo
from . import modem as omodem
from . import ofdm as oofdm
from . import philox


class LegacyRng:
    """Sequential draws from one legacy MT19937 stream, as the reference does."""
    legacy = True

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def symbols(self, n, M):
        return self.rs.randint(0, M, n)

    def cn(self, stream, *shape):
        # util/misc.py:354-355: all real parts are drawn before all imaginary parts
        re = self.rs.randn(*shape)
        im = self.rs.randn(*shape)
        return (1.0 / math.sqrt(2.0)) * (re + 1j * im)

    def uniform(self, *shape):
        return self.rs.rand(*shape)


class LegacyRng3(LegacyRng):
    """Config 5 draws from three legacy streams: the channel's and the noise's own RandomState
    (multiuser.py:670-709) and the global one for the data -- all seeded alike by the harness."""

    def __init__(self, seed):
        super().__init__(seed)
        self.by_stream = {philox.STREAM_CHAN: np.random.RandomState(seed),
                          philox.STREAM_NOISE: np.random.RandomState(seed),
                          3: np.random.RandomState(seed)}       # the iterative solvers' own RandomState

    def cn(self, stream, *shape):
        rs = self.by_stream[stream]
        re = rs.randn(*shape)
        im = rs.randn(*shape)
        return (1.0 / math.sqrt(2.0)) * (re + 1j * im)


class PhiloxRng:
    """Position-addressed draws; each stream keeps its own running position."""
    legacy = False

    def __init__(self, seed, realization):
        self.seed, self.r = seed, realization
        self.pos = {}

    def _advance(self, stream, n):
        p = self.pos.get(stream, 0)
        self.pos[stream] = p + n
        return p

    def symbols(self, n, M):
        return philox.symbols(self.seed, self.r, n, M, offset=self._advance(philox.STREAM_DATA, n))

    def cn(self, stream, *shape):
        n = int(np.prod(shape))
        z = philox.cnormal(self.seed, self.r, n, stream, offset=self._advance(stream, n))
        return z.reshape(shape)

    def uniform(self, *shape):
        n = int(np.prod(shape))
        u = philox.uniforms(self.seed, self.r, n, offset=self._advance(philox.STREAM_PHASE, n))
        return u.reshape(shape)


def constellation(mod, M):
    if mod == 'qam':
        return omodem.qam_constellation(M)
    if mod == 'psk':
        return omodem.psk_constellation(M)
    if mod == 'qpsk':
        return omodem.psk_constellation(4, math.pi / 4.0)
    if mod == 'bpsk':
        return omodem.bpsk_constellation()
    raise ValueError(mod)


def _counts(out, idx, dec, M):
    out['decisions'] = dec
    out['symbol_errors'] = omodem.count_symbol_errors(idx, dec)
    out['bit_errors'] = int(omodem.count_bit_errors(idx, dec))
    out['num_symbols'] = int(idx.size)
    out['num_bits'] = int(idx.size) * omodem.level2bits(M)
    return out


def chain_awgn(rng, mod='qam', M=16, N=10000, snr_db=10.0):
    """C1: idx -> modulate -> + sqrt(1/snr) * randn_c -> demodulate -> counts."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    idx = rng.symbols(N, M)
    tx = omodem.modulate(table, idx)
    noise = rng.cn(philox.STREAM_NOISE, N)
    rx = tx + math.sqrt(noise_var) * noise
    dec = omodem.demodulate(table, rx)
    return _counts(dict(table=table, idx=idx, tx=tx, noise=noise, rx=rx, noise_var=noise_var),
                   idx, dec, M)


def _jakes_phases(rng, L, shape):
    """phi then psi, each 2*pi*rand(L, *shape, 1) (fading_generators.py:421-425)."""
    phi = 2 * np.pi * rng.uniform(L, *shape, 1)
    psi = 2 * np.pi * rng.uniform(L, *shape, 1)
    return phi, psi


def chain_flat_jakes(rng, mod='qam', M=64, N=100000, snr_db=20.0, Fd=100.0, Ts=1e-3, L=8):
    """C2: flat fading SuChannel(JakesSampleGenerator(Fd, Ts, L)); y = h*s + n; equalise y/h."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    idx = rng.symbols(N, M)
    tx = omodem.modulate(table, idx)
    if rng.legacy:
        # Jakes ctor: phi/psi of shape (L, 1) used for one discarded sample at t = 0
        # (fading_generators.py:348-351,413-414)
        rng.uniform(L, 1)
        rng.uniform(L, 1)
    # TdlChannel ctor sets the generator shape to (num_taps,) -> re-draw (fading.py:796-798)
    phi, psi = _jakes_phases(rng, L, (1,))
    t, _ = och.jakes_time_axis(Ts, Ts, N)            # ctor consumed t = 0, so t0 = Ts
    h = och.jakes_samples(phi, psi, Fd, t)            # [1, N]
    p_lin, d_idx = och.discretize_profile(np.zeros(1), np.zeros(1), Ts)
    taps = och.tdl_taps(h, p_lin)
    faded = och.tdl_apply(tx, taps, d_idx)
    noise = rng.cn(philox.STREAM_NOISE, N)
    rx = faded + math.sqrt(noise_var) * noise
    eq = rx / taps[0]
    dec = omodem.demodulate(table, eq)
    return _counts(dict(table=table, idx=idx, tx=tx, phi=phi, psi=psi, t=t, h=taps, faded=faded,
                        noise=noise, rx=rx, eq=eq, noise_var=noise_var), idx, dec, M)


def chain_flat_rayleigh(rng, mod='qam', M=16, N=1000, snr_db=15.0, form='suchannel'):
    """C2b: flat i.i.d. Rayleigh fading, y = h*s + n, equalise y/h, h ~ CN(0, 1) per sample.

    form='suchannel'  SuChannel(RayleighSampleGenerator()) (fading_generators.py:208-249, singleuser.py:48-81):
                      draws idx; generator ctor randn_c() (:228-229, one discarded sample); the TdlChannel ctor
                      only sets the shape to (num_taps,) (fading.py:796-798; unlike Jakes the Rayleigh setter does
                      not re-draw, fading_generators.py:138-154); corrupt_data -> randn_c(1, N) = h
                      (fading_generators.py:244-249, fading.py:940); then the noise randn_c(N).
    form='notebook'   notebooks/Transmission_with_Rayleigh_and_AWGN_channels.ipynb cell 8
                      (RayleighOrAwgnSimulator._run_simulation): idx; noise randn_c(N); h = randn_c(N).
    Under the Philox contract both forms address the same positions (h = CHAN samples 0..N-1, noise = NOISE
    samples 0..N-1), so they are the same chain; under a sequential legacy stream the draw order differs."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    idx = rng.symbols(N, M)
    tx = omodem.modulate(table, idx)
    if form == 'notebook':
        noise = rng.cn(philox.STREAM_NOISE, N)
        h = rng.cn(philox.STREAM_CHAN, N).reshape(1, N)
        faded = h[0] * tx
    else:
        if rng.legacy:
            rng.cn(philox.STREAM_CHAN)           # RayleighSampleGenerator ctor: one sample, shape None
        h = rng.cn(philox.STREAM_CHAN, 1, N)     # generate_more_samples(N) with shape (1,)
        p_lin, d_idx = och.discretize_profile(np.zeros(1), np.zeros(1), 1.0)
        taps = och.tdl_taps(h, p_lin)
        faded = och.tdl_apply(tx, taps, d_idx)
        h = taps
        noise = rng.cn(philox.STREAM_NOISE, N)
    rx = faded + math.sqrt(noise_var) * noise
    eq = rx / h[0]
    dec = omodem.demodulate(table, eq)
    return _counts(dict(table=table, idx=idx, tx=tx, h=h, faded=faded, noise=noise, rx=rx, eq=eq,
                        noise_var=noise_var), idx, dec, M)


def chain_ofdm_tdl(rng, mod='qpsk', M=4, fft_size=1024, cp_size=16, num_used=None, n_ofdm_sym=1,
                   snr_db=20.0, Fd=10.0, Ts=1.0 / (15e3 * 1024), L=8,
                   tap_powers_dB=(0.0, -3.0, -6.0, -9.0, -12.0), tap_delays_samples=(0, 1, 2, 3, 4), linear_mean=False):
    """C3: OFDM over a time-varying Jakes TDL channel with a one-tap equaliser.
    linear_mean: the equaliser's response as the DFT of the per-symbol mean taps (oofdm.onetap_equalize_fast, equal to the literal
    mean of per-sample DFTs by linearity, 10 x cheaper: the deep GPU parity tests use it)."""
    table = constellation(mod, M)
    num_used = oofdm.check_params(fft_size, cp_size, num_used)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    n_data = num_used * n_ofdm_sym
    idx = rng.symbols(n_data, M)
    sym = omodem.modulate(table, idx)
    tx = oofdm.modulate(sym, fft_size, cp_size, num_used)
    p_lin, d_idx = och.discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                          np.asarray(tap_delays_samples, dtype=float) * Ts, Ts)
    ntaps = len(d_idx)
    if rng.legacy:
        rng.uniform(L, 1)
        rng.uniform(L, 1)
    phi, psi = _jakes_phases(rng, L, (ntaps,))
    t, _ = och.jakes_time_axis(Ts, Ts, tx.size)
    fading = och.jakes_samples(phi, psi, Fd, t)       # [taps, n]
    taps = och.tdl_taps(fading, p_lin)
    faded = och.tdl_apply(tx, taps, d_idx)            # n + max_delay samples
    noise = rng.cn(philox.STREAM_NOISE, faded.size)
    rx = faded + math.sqrt(noise_var) * noise
    demod = oofdm.demodulate(rx[:tx.size].copy(), fft_size, cp_size, num_used)
    eq = (oofdm.onetap_equalize_fast if linear_mean else oofdm.onetap_equalize)(demod, taps, d_idx, fft_size, cp_size, num_used)
    dec = omodem.demodulate(table, eq)
    return _counts(dict(table=table, idx=idx, sym=sym, tx=tx, phi=phi, psi=psi, t=t, taps=taps,
                        tap_powers_linear=p_lin, delay_indexes=d_idx, faded=faded, noise=noise,
                        rx=rx, demod=demod, eq=eq, noise_var=noise_var), idx, dec, M)


def chain_mimo_ofdm(rng, mod='qam', M=64, nt=4, nr=4, fft_size=1024, cp_size=16, num_used=None,
                    n_ofdm_sym=1, snr_db=25.0, mmse=True):
    """C4: flat H = randn_c(Nr, Nt) per realization, Blast encode, per-antenna OFDM,
    R = H T + noise, per-antenna OFDM demodulate, Blast MMSE (or ZF) decode."""
    table = constellation(mod, M)
    num_used = oofdm.check_params(fft_size, cp_size, num_used)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    H = rng.cn(philox.STREAM_CHAN, nr, nt)
    n_data = nt * num_used * n_ofdm_sym
    idx = rng.symbols(n_data, M)
    sym = omodem.modulate(table, idx)
    X = omimo.blast_encode(sym, nt)                                        # [nt, n/nt]
    T = np.stack([oofdm.modulate(X[a], fft_size, cp_size, num_used) for a in range(nt)])
    noise = rng.cn(philox.STREAM_NOISE, nr, T.shape[1])
    R = H @ T + math.sqrt(noise_var) * noise
    Y = np.stack([oofdm.demodulate(R[a].copy(), fft_size, cp_size, num_used) for a in range(nr)])
    nv_filter = noise_var if mmse else 0.0
    G = omimo.blast_receive_filter(H, nv_filter)
    est = (G @ Y).reshape(-1, order='F')
    dec = omodem.demodulate(table, est)
    return _counts(dict(table=table, H=H, idx=idx, sym=sym, X=X, T=T, noise=noise, R=R, Y=Y, G=G,
                        est=est, noise_var=noise_var), idx, dec, M)


def chain_ia(rng, mod='qam', M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0):
    """C5: apps/ia/simulate_ia.py:94-245 with ClosedFormIASolver(use_best_init=True)."""
    assert K == 3
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    big_H = rng.cn(philox.STREAM_CHAN, K * nr, K * nt)
    H = oia.split_blocks(big_H, K, nr, nt)
    F, U, cap, sinr = oia.closed_form_solve(H, Ns, noise_var)
    if rng.legacy:
        idx = rng.rs.randint(0, M, [K * Ns, NSymbs])
    else:
        idx = rng.symbols(K * Ns * NSymbs, M).reshape(K * Ns, NSymbs)
    sym = omodem.modulate(table, idx)
    X = np.vstack([F[k] @ sym[k * Ns:(k + 1) * Ns] for k in range(K)])
    noise = rng.cn(philox.STREAM_NOISE, K * nr, NSymbs)
    Y = oia.mu_corrupt(big_H, X, noise, noise_var)
    est = np.vstack([U[k] @ Y[k * nr:(k + 1) * nr] for k in range(K)])
    dec = omodem.demodulate(table, est)
    out = dict(table=table, big_H=big_H, idx=idx, noise=noise, est=est, noise_var=noise_var, sum_capacity=cap,
               F=np.stack([f.reshape(-1) for f in F]) if Ns == 1 else None,
               U=np.stack([u.reshape(-1) for u in U]) if Ns == 1 else None,
               sinr=np.concatenate(sinr))
    return _counts(out, idx, dec, M)


def chain_mimo_scheme(rng, scheme='blast', mod='qam', M=16, nt=2, nr=2, NSymbs=200, snr_db=15.0):
    """apps/mimo/simulate_mimo.py:68-142: flat channel randn_c(Nr, Nt) per realization, one of the six MIMO
    schemes (Alamouti / Blast / MRC / MRT / SVDMimo / GMDMimo -- set_channel_matrix only, i.e. zero forcing),
    NSymbs symbols per layer, single carrier."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    H = rng.cn(philox.STREAM_CHAN, nr, nt)
    layers = {'blast': nt, 'mrc': nt, 'svd': nt, 'gmd': nt, 'alamouti': 1, 'mrt': 1}[scheme]
    idx = rng.symbols(NSymbs * layers, M)
    sym = omodem.modulate(table, idx)
    if scheme in ('blast', 'mrc'):
        X = omimo.blast_encode(sym, nt)
    elif scheme == 'alamouti':
        X = omimo.alamouti_encode(sym)
    elif scheme == 'mrt':
        X = omimo.mrt_encode(sym, H)
    elif scheme == 'svd':
        X = omimo.svd_encode(sym, H)
    else:
        X = omimo.gmd_encode(sym, H)
    noise = rng.cn(philox.STREAM_NOISE, nr, NSymbs)
    Y = H @ X + math.sqrt(noise_var) * noise
    if scheme in ('blast', 'mrc'):
        est = omimo.blast_decode(Y, H, 0.0)
    elif scheme == 'alamouti':
        est = omimo.alamouti_decode(Y, H)
    elif scheme == 'mrt':
        est = omimo.mrt_decode(Y, H)
    elif scheme == 'svd':
        est = omimo.svd_decode(Y, H)
    else:
        est = omimo.gmd_decode(Y, H, 0.0)
    dec = omodem.demodulate(table, est)
    out = dict(table=table, H=H, idx=idx, noise=noise, est=est, noise_var=noise_var)
    if scheme in ('svd', 'gmd'):      # the basis-dependent pair (mimo.py:846-890, 965-1011), as minted into the fixture
        out['W'], out['G_H'] = omimo.scheme_filters(scheme, H)
    return _counts(out, idx, dec, M)


def apply_pathloss(big_H, K, pathloss):
    """multiuser.py:256-292: block (rx k, tx l) of the channel scaled by sqrt(pathloss[k, l])."""
    if pathloss is None:
        return big_H
    pl = np.asarray(pathloss, dtype=float)
    nr, nt = big_H.shape[0] // K, big_H.shape[1] // K
    return big_H * np.kron(np.sqrt(pl), np.ones((nr, nt)))


def chain_bd(rng, mod='psk', M=4, K=3, nr=2, NSymbs=500, iPu=1.0, noise_var=0.03, bd_noise_var=1e-50,
             pathloss=None, waterfill=True, canonical=False):
    """SURVEY.md section 8(f).3 tail: apps/comp_BD/simulate_comp_simple.py:95-140 without external
    interference -- K cells of nr x nr antennas transmit jointly, block diagonalisation of the (K nr) x
    (K nr) channel (path loss applied per block), zero-forcing pinv(newH) at the receivers.
    canonical=True uses the kernels' formulation (canonical singular-vector phases) instead of numpy's SVD:
    the same precoder up to one phase per stream, hence the same statistics but not the same decisions."""
    table = constellation(mod, M)
    n = K * nr
    big_H = apply_pathloss(rng.cn(philox.STREAM_CHAN, n, n), K, pathloss)
    if rng.legacy:
        idx = rng.rs.randint(0, M, [n, NSymbs])
    else:
        idx = rng.symbols(n * NSymbs, M).reshape(n, NSymbs)
    sym = omodem.modulate(table, idx)
    if canonical:
        newH, Ms, W = obd.block_diagonalize_closed(big_H, K, iPu, bd_noise_var, waterfill)
    else:
        newH, Ms = (obd.block_diagonalize(big_H, K, iPu, bd_noise_var) if waterfill
                    else obd.block_diagonalize_no_waterfilling(big_H, K, iPu))
        W = obd.calc_receive_filter(newH)
    noise = rng.cn(philox.STREAM_NOISE, n, NSymbs)
    Y = oia.mu_corrupt(big_H, Ms @ sym, noise, noise_var)
    est = W @ Y
    dec = omodem.demodulate(table, est)
    return _counts(dict(table=table, big_H=big_H, idx=idx, noise=noise, est=est, noise_var=noise_var, Ms=Ms,
                        newH=newH, W=W), idx, dec, M)


STREAM_INIT = 3    # the solver's own RandomState (iabase.py:95); shares the PHASE stream id, unused in config 5


def chain_ia_iterative(rng, algo='alt_min', mod='qam', M=16, K=3, nr=2, nt=2, Ns=1, NSymbs=200, snr_db=20.0,
                       max_iterations=50, relative_factor=1e-6, initialize_with='random'):
    """apps/ia/simulate_ia.py:94-245 with an iterative solver
    (AlternatingMinIASolver / MinLeakageIASolver / MaxSinrIASolver, initialize_with='random')."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    big_H = rng.cn(philox.STREAM_CHAN, K * nr, K * nt)
    H = oia.split_blocks(big_H, K, nr, nt)
    F_init = []
    if initialize_with in ('closed_form', 'svd'):
        F_init = [np.zeros((nt, Ns), dtype=complex)] * K       # shape only: no random start is drawn
    else:
        # randomizeF (iabase.py:538-540): normalized(randn_c_RS(rs, Nt, Ns)) -- drawn by the solver itself
        # ('random') or by the alternating-minimisation solver it starts from ('alt_min'), each from the
        # beginning of its own RandomState
        for k in range(K):
            f = rng.cn(STREAM_INIT, nt, Ns)
            F_init.append(f / np.linalg.norm(f, "fro"))
    F, U, cap, sinr, runned = oia.iterative_solve(algo, H, F_init, noise_var, max_iterations, relative_factor,
                                                  initialize_with)
    if rng.legacy:
        idx = rng.rs.randint(0, M, [K * Ns, NSymbs])
    else:
        idx = rng.symbols(K * Ns * NSymbs, M).reshape(K * Ns, NSymbs)
    sym = omodem.modulate(table, idx)
    X = np.vstack([F[k] @ sym[k * Ns:(k + 1) * Ns] for k in range(K)])
    noise = rng.cn(philox.STREAM_NOISE, K * nr, NSymbs)
    Y = oia.mu_corrupt(big_H, X, noise, noise_var)
    est = np.vstack([U[k] @ Y[k * nr:(k + 1) * nr] for k in range(K)])
    dec = omodem.demodulate(table, est)
    out = dict(table=table, big_H=big_H, idx=idx, noise=noise, est=est, noise_var=noise_var, sum_capacity=cap,
               F_init=np.stack([f.reshape(-1) for f in F_init]) if Ns == 1 else None,
               F=np.stack([f.reshape(-1) for f in F]) if Ns == 1 else None,
               U=np.stack([u.reshape(-1) for u in U]) if Ns == 1 else None,
               sinr=np.concatenate(sinr), runned_iterations=runned)
    return _counts(out, idx, dec, M)


def chain_ia_general(rng, algo, mod, M, K, nr, nt, Ns, NSymbs, snr_db, max_iterations, relative_factor,
                      initialize_with="random", select=None):
    """SURVEY.md section 8(f).3 tail: iterative IA on general geometries (per-user stream counts, 'svd' start,
    GreedStreamIASolver / BruteForceStreamIASolver; ia/algorithms.py:802-883, 885-1507, 1853-2260) and the link run
    with the solution found (apps/ia/simulate_ia.py:94-245).  Legacy streams only (fixture pinning)."""
    table = constellation(mod, M)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    big_H = rng.cn(philox.STREAM_CHAN, K * nr, K * nt)
    H = oia.split_blocks(big_H, K, nr, nt)
    Ns_l = [int(Ns)] * K if np.isscalar(Ns) else [int(n) for n in Ns]
    F_init = None
    if initialize_with == "random" and select != "brute":
        F_init = []
        for k in range(K):
            f = rng.cn(STREAM_INIT, nt, Ns_l[k])            # the solver's own RandomState
            F_init.append(f / np.linalg.norm(f, "fro"))
    sol = oia.general_solve(algo, H, Ns_l, noise_var, max_iterations, relative_factor, F_init, select)
    Ns_fin = sol["Ns"]
    idx = rng.rs.randint(0, M, [int(np.sum(Ns_fin)), NSymbs])
    sym = omodem.modulate(table, idx)
    cum = np.cumsum(Ns_fin)
    tx = np.split(sym, cum[:-1])
    X = np.vstack([sol["F"][k] @ tx[k] for k in range(K)])
    noise = rng.cn(philox.STREAM_NOISE, K * nr, NSymbs)
    Y = oia.mu_corrupt(big_H, X, noise, noise_var)
    est = np.vstack([sol["U"][k] @ Y[k * nr:(k + 1) * nr] for k in range(K)])
    dec = omodem.demodulate(table, est)
    pad = lambda mats, r, c: np.stack([np.pad(np.asarray(a), ((0, r - a.shape[0]), (0, c - a.shape[1]))) for a in mats])
    out = dict(table=table, big_H=big_H, idx=idx, noise=noise, est=est, decisions=dec, noise_var=noise_var,
               sum_capacity=sol["cap"], Ns_final=np.array(Ns_fin, dtype=int),
               PF=np.stack([f @ f.conj().T for f in sol["F"]]), PU=np.stack([u.conj().T @ u for u in sol["U"]]),
               sinr=np.concatenate([np.asarray(s, dtype=float) for s in sol["sinr"]]),
               runned_iterations=int(sol["runned"]), symbol_errors=omodem.count_symbol_errors(idx, dec), bit_errors=int(omodem.count_bit_errors(idx, dec)),
               num_symbols=int(idx.size), num_bits=int(idx.size) * omodem.level2bits(M))
    if F_init is not None:
        D = 4 if max(nr, nt) <= 4 else 6          # the solver's two matrix capacities (csrc/kernels_ia_general.hip)
        out["F_init"] = pad(F_init, D, D)
    if select == "brute":
        out["every_sum_capacity"] = np.array(sol["every_sum_capacity"], dtype=float)
        out["stream_combinations"] = np.array(sol["stream_combinations"], dtype=np.int64)
    return out


def chain_mimo_ofdm_tdl(rng, mod='qam', M=16, nt=2, nr=2, fft_size=64, cp_size=16, num_used=None, n_ofdm_sym=2,
                        snr_db=20.0, Fd=50.0, Ts=1e-6, L=8, tap_powers_dB=(0.0, -4.0, -9.0),
                        tap_delays_samples=(0, 2, 5), mmse=True, linear_mean=False):
    """spatial multiplexing over a frequency-selective MIMO TDL channel
    (TdlMimoChannel, fading.py:1290-1333 + the MIMO branch of corrupt_data :1107-1117), per-antenna OFDM
    and one MMSE receive filter per subcarrier built from the per-symbol mean frequency response
    (get_freq_response :513-536; Blast._calc_receive_filter mimo.py:577-607 on every used bin)."""
    table = constellation(mod, M)
    used = oofdm.check_params(fft_size, cp_size, num_used)
    noise_var = 1.0 / float(omodem.dB2Linear(snr_db))
    idx = rng.symbols(nt * used * n_ofdm_sym, M)
    sym = omodem.modulate(table, idx)
    X = omimo.blast_encode(sym, nt)
    T = np.stack([oofdm.modulate(X[a], fft_size, cp_size, used) for a in range(nt)])
    p_lin, d_idx = och.discretize_profile(np.asarray(tap_powers_dB, dtype=float),
                                          np.asarray(tap_delays_samples, dtype=float) * Ts, Ts)
    S = len(d_idx)
    if rng.legacy:
        rng.uniform(L, nr, nt, 1)      # Jakes ctor with shape (nr, nt): one discarded sample
        rng.uniform(L, nr, nt, 1)
    phi, psi = _jakes_phases(rng, L, (S, nr, nt))
    n = T.shape[1]
    t, _ = och.jakes_time_axis(Ts, Ts, n)
    fading = och.jakes_samples(phi, psi, Fd, t)                  # [S, nr, nt, n]
    taps = och.tdl_taps(fading, p_lin)
    faded = och.tdl_apply_mimo(T, taps, d_idx)                   # [nr, n + dmax]
    noise = rng.cn(philox.STREAM_NOISE, nr, faded.shape[1])
    R = faded + math.sqrt(noise_var) * noise
    Y = np.stack([oofdm.demodulate(R[r, :n].copy(), fft_size, cp_size, used) for r in range(nr)])   # [nr, ns]
    # linear_mean: the DFT of the per-symbol mean taps instead of the mean of per-sample DFTs (equal by linearity, 30 x cheaper:
    # och.mean_freq_response_linear; the deep GPU parity tests use it)
    Hm = (och.mean_freq_response_linear if linear_mean else och.mean_freq_response)(taps, d_idx, fft_size, cp_size, n_ofdm_sym)
    Hu = Hm[:, oofdm.used_subcarrier_indexes(fft_size, used)].reshape(-1, nr, nt)  # [ns, nr, nt]
    G = np.stack([omimo.blast_receive_filter(Hu[c], noise_var if mmse else 0.0) for c in range(Hu.shape[0])])
    est = np.einsum('car,rc->ca', G, Y).reshape(-1)               # est[c*nt + a]
    dec = omodem.demodulate(table, est)
    return _counts(dict(table=table, idx=idx, T=T, phi=phi, psi=psi, taps=taps, delay_indexes=d_idx,
                        tap_powers_linear=p_lin, faded=faded, noise=noise, Y=Y, Hu=Hu, G=G, est=est,
                        noise_var=noise_var), idx, dec, M)
