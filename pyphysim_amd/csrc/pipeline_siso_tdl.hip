// pipeline_siso_tdl.hip -- config 3 (SISO OFDM over a time-varying Jakes TDL channel, one-tap equaliser),
// NB = 4 realizations per workgroup pass.
//
// The single-realization kernel (pipelines.hip: k_run_ofdm_tdl) is latency bound: 256 threads share one
// 1024-point transform, one butterfly per thread per stage, a dozen barriers per realization and eight rays
// per tap and sample.  Here the four "antenna" rows of the MIMO kernels carry four independent realizations:
// the IFFT / FFT stages run four butterflies per thread with shared twiddles, every barrier serves four
// realizations, and -- as in pipeline_mimo_tdl.hip -- each tap is a short polynomial in the sample index around
// the middle of the OFDM symbol (order K chosen by the host from the Doppler phase across half a symbol; beyond
// kMaxOrder the launcher falls back to the single-realization kernel, which rotates the rays sample by sample).
//
// Reference path (restated by oracle/chains.py::chain_ofdm_tdl): notebooks/TDL_and_OFDM.ipynb
// OfdmTdlSimulator._run_simulation; modulators/ofdm.py:394-466,515-552; channels/fading.py:1046-1090;
// channels/fading_generators.py:421-425,459-467,519-522.
// Draw ledger per realization (mcle-philox-v1): DATA symbol n = os*U + d; PHASE phi = uniform l*S + s,
// psi = uniform L*S + l*S + s; NOISE sample j of the faded stream.
#include <type_traits>

#include "fft.hpp"
#include "jakes.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "totals.hpp"

namespace mcle {

constexpr int kSisoMaxOrder = 12;

struct SisoTdlParams {
    int cp, num_used, n_ofdm_sym;
    int n_taps, L, K, dmax;
    int x_elems;                     // complex elements of the sample buffer (>= NB*N; also holds the ray scratch)
    double noise_var, Fd, Ts, dt;
    double tap_amp[MCLE_MAX_TAPS];   // sqrt(p_s / L)
    int tap_delay[MCLE_MAX_TAPS];
    double mom[kSisoMaxOrder + 1];   // mean over the symbol's N+cp samples of x^m, x = j - (N+cp-1)/2
};

template <typename T, int N, int NB>
__global__ __launch_bounds__(kPipeBlock, sizeof(T) == 4 ? 3 : 1) void k_run_ofdm_tdl_batch(
    SisoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first, uint64_t count,
    const cx<T>* __restrict__ g_tw, mcle_counters* counters, uint32_t* __restrict__ sym_out,
    uint32_t* __restrict__ bit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = pp.n_taps, L = pp.L, K = pp.K, dmax = pp.dmax;
    const int PS = S * NB;                              // fading processes of a pass: slot a, tap s -> a*S + s
    cx<T>* s_x = reinterpret_cast<cx<T>*>(smem);       // [NB][N] (+ slack for the ray scratch of small FFTs)
    cx<T>* s_tw = s_x + pp.x_elems;                     // [N]
    cx<T>* s_coef = s_tw + N;                           // [PS][K+1]
    cx<T>* s_mean = s_coef + PS * (K + 1);              // [PS]
    cx<T>* s_tail = s_mean + PS;                        // [2][NB][dmax] last samples of the previous symbol
    cx<T>* s_table = s_tail + 2 * NB * (dmax > 0 ? dmax : 1);   // [kMaxTable]
    unsigned* s_red = reinterpret_cast<unsigned*>(s_table + kMaxTable);   // [2*NB*4] per-wave partials
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_red + 2 * NB * (kPipeBlock / 64));
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [NB][num_used]

    const int tid0 = threadIdx.x;
    for (int k = tid0; k < N; k += kPipeBlock) s_tw[k] = g_tw[k];
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const double xc = 0.5 * (double)(W - 1);            // centre of the symbol in local sample units
    __shared__ WgTotals totals;
    if (tid0 == 0) wg_zero(totals);

    // one butterfly position per thread and stage (N = 4 * threads, radix-4 only): its twiddles live in registers
    constexpr bool kTwRegs = (N == 4 * kPipeBlock) && !FftShape<N>::HAS2 && sizeof(T) == 4;
    cx<T> twr[FftShape<N>::N4][3];
    if constexpr (kTwRegs) {
        __syncthreads();
        fft_twiddle_regs<T, N, kPipeBlock>(s_tw, twr);
    }
    const uint64_t n_pass = (count + NB - 1) / NB;
    for (uint64_t ps = blockIdx.x; ps < n_pass; ps += gridDim.x) {
        const uint64_t base = ps * NB;                  // slot a carries realization base + a (idle past `count`)
        unsigned se[NB], be[NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) se[a] = be[a] = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            const int tid = opaque(tid0);
            __syncthreads();
            // ---- tap polynomials of this symbol: one ray per thread, then one (process, order) per thread ----
            {
                const double two_pi = 6.283185307179586476925286766559;
                const double tc = pp.Ts + pp.dt * ((double)sym0 + xc);
                T* s_ray = reinterpret_cast<T*>(s_x);                    // [PS*L][3] = {re, im, theta}
                for (int q = tid; q < PS * L; q += kPipeBlock) {
                    const int a = q / (S * L), rq = q - a * (S * L);      // rq = l*S + s: PHASE-stream index of phi
                    const int l = rq / S, s = rq - l * S;
                    const Rng rng(seed, first + base + a);
                    const double psi_t = uniform_at(rng, STREAM_PHASE, (uint64_t)L * S + rq);
                    const double w = pp.Fd * cospi(2.0 * uniform_at(rng, STREAM_PHASE, (uint64_t)rq));   // Hz
                    const double ph = fma(w, tc, psi_t);                  // turns
                    const double fr = ph - floor(ph);
                    T er, ei;
                    if constexpr (sizeof(T) == 8) {
                        double sn, cs;
                        sincos(two_pi * fr, &sn, &cs);
                        er = cs;
                        ei = sn;
                    } else {
                        er = __builtin_amdgcn_cosf((float)fr);
                        ei = __builtin_amdgcn_sinf((float)fr);
                    }
                    T* o = s_ray + 3 * ((a * S + s) * L + l);
                    o[0] = er;
                    o[1] = ei;
                    o[2] = (T)(two_pi * w * pp.dt);                       // rad per sample
                }
                __syncthreads();
                for (int q = tid; q < PS * (K + 1); q += kPipeBlock) {
                    const int p = q / (K + 1), m = q - p * (K + 1);
                    T inv_fact = 1;
                    for (int i = 2; i <= m; ++i) inv_fact /= (T)i;
                    T ar = 0, ai = 0;
                    for (int l = 0; l < L; ++l) {
                        const T* o = s_ray + 3 * (p * L + l);
                        T pw = inv_fact;
                        for (int i = 0; i < m; ++i) pw *= o[2];
                        ar += o[0] * pw;
                        ai += o[1] * pw;
                    }
                    T cr, ci;                                             // times j^m
                    switch (m & 3) {
                        case 0: cr = ar; ci = ai; break;
                        case 1: cr = -ai; ci = ar; break;
                        case 2: cr = -ar; ci = -ai; break;
                        default: cr = ai; ci = -ar; break;
                    }
                    const T amp = (T)pp.tap_amp[p % S];
                    s_coef[q] = mk<T>(amp * cr, amp * ci);
                }
                __syncthreads();
                for (int p = tid; p < PS; p += kPipeBlock) {
                    T mr = 0, mi = 0;
                    for (int m = 0; m <= K; ++m) {
                        const cx<T> c = s_coef[p * (K + 1) + m];
                        mr += c.x * (T)pp.mom[m];
                        mi += c.y * (T)pp.mom[m];
                    }
                    s_mean[p] = mk<T>(mr, mi);
                }
                __syncthreads();   // the ray scratch is dead: the sample buffer may be refilled
            }
            // ---- transmit: symbols -> bins, every slot from its own DATA stream ----
            if (U != N) {
                for (int p = tid; p < NB * N; p += kPipeBlock) s_x[p] = mk<T>(0, 0);
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * U, n_last = n_first + U;
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const Rng rng(seed, first + base + a);
                for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += kPipeBlock) {
                    const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint64_t n = (blk << 4) + j;
                        if (n >= n_first && n < n_last) {
                            const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                            const int d = (int)(n - n_first);
                            s_idx[a * U + d] = (unsigned char)tx;
                            s_x[a * N + lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(d, N, U)))] = cscale(s_table[tx], tx_scale);
                        }
                    }
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dit_r<T, N, true, kPipeBlock, true, true>(s_x, NB, N, twr);
            else
                fft_dit<T, N, true, kPipeBlock, true, true>(s_x, NB, N, s_tw);   // bins scattered digit-reversed -> time samples in natural order
            auto time_sample = [&](int a, int i) -> cx<T> {                   // IFFT output i of slot a
                return s_x[a * N + lds_swz<true>(i & (N - 1))];
            };
            const int tid_c = opaque(tid0);
            cx<T>* tail_prev = s_tail + (size_t)(os & 1) * NB * dmax;
            cx<T>* tail_next = s_tail + (size_t)((os + 1) & 1) * NB * dmax;
            if (os + 1 < pp.n_ofdm_sym)
                for (int q = tid_c; q < NB * dmax; q += kPipeBlock) {
                    const int a = q / dmax, i = q - a * dmax;
                    tail_next[q] = time_sample(a, N - dmax + i);
                }
            // ---- channel: y[a][m] = sum_s g[a][s](j) T[a][j],  j = cp + m - d_s (input sample) ----
            constexpr int PAIRS = (N / 2 + kPipeBlock - 1) / kPipeBlock;   // sample pairs per thread
            cx<T> y[NB][PAIRS][2];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int k = 0; k < PAIRS; ++k) y[a][k][0] = y[a][k][1] = mk<T>(0, 0);
            auto channel = [&](auto fast_tag, auto k_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                constexpr int KT = decltype(k_tag)::value;
                for (int s = 0; s < S; ++s) {
                    const int d = pp.tap_delay[s];
                    int pos[PAIRS][2];      // >= 0: offset in a slot's sample row; -1: zero; <= -2: tail slot -2-i
                    T xx[PAIRS][2];
#pragma unroll
                    for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int m = 2 * (tid_c + kPipeBlock * k) + e;
                            const int q = cp + m - d;                // local index of the input sample
                            xx[k][e] = (T)((double)q - xc);
                            if (FAST || (m < N && q >= 0))
                                pos[k][e] = lds_swz<true>((m - d + N) & (N - 1));
                            else if (m >= N)
                                pos[k][e] = -1;
                            else
                                pos[k][e] = os > 0 ? -2 - (dmax + q) : -1;   // sample W + q of the previous symbol
                        }
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const cx<T>* c = s_coef + (a * S + s) * (K + 1);
                        cx<T> cc[KT > 0 ? KT + 1 : 1];
                        if constexpr (KT > 0) {
#pragma unroll
                            for (int m = 0; m <= KT; ++m) cc[m] = c[m];
                        }
#pragma unroll
                        for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int ps_ = pos[k][e];
                                cx<T> xv;
                                if (FAST)
                                    xv = s_x[a * N + ps_];
                                else
                                    xv = ps_ >= 0 ? s_x[a * N + ps_]
                                                  : (ps_ == -1 ? mk<T>(0, 0) : tail_prev[a * dmax + (-2 - ps_)]);
                                cx<T> g;
                                if constexpr (KT > 0) {
                                    g = cc[KT];
#pragma unroll
                                    for (int m = KT - 1; m >= 0; --m) {
                                        g.x = fma(g.x, xx[k][e], cc[m].x);
                                        g.y = fma(g.y, xx[k][e], cc[m].y);
                                    }
                                } else {
                                    g = c[K];
                                    for (int m = K - 1; m >= 0; --m) {
                                        const cx<T> cm = c[m];
                                        g.x = fma(g.x, xx[k][e], cm.x);
                                        g.y = fma(g.y, xx[k][e], cm.y);
                                    }
                                }
                                y[a][k][e] = cfma(g, xv, y[a][k][e]);
                            }
                    }
                }
            };
            {
                const bool fast = cp >= dmax && (N / 2) % kPipeBlock == 0;
                typedef std::integral_constant<int, 0> k_any;
                typedef std::integral_constant<int, 2> k_two;
                if (fast) {
                    if (K == 2) channel(std::true_type{}, k_two{});
                    else channel(std::true_type{}, k_any{});
                } else {
                    if (K == 2) channel(std::false_type{}, k_two{});
                    else channel(std::false_type{}, k_any{});
                }
            }
            // noise of the samples that survive CP removal: sample sym0 + cp + m of every slot's own NOISE stream
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                const int m0 = 2 * (tid_c + kPipeBlock * k);
                if (m0 < N) {
                    const uint64_t i0 = sym0 + cp + m0;
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const Rng rng(seed, first + base + a);
                        cx<T> z0, z1;
                        if ((i0 & 1) == 0) {
                            cn_pair<T>(rng, STREAM_NOISE, (uint32_t)(i0 >> 1), sigma, z0, z1);
                        } else {
                            z0 = cn_sample<T>(rng, STREAM_NOISE, i0, sigma);
                            z1 = cn_sample<T>(rng, STREAM_NOISE, i0 + 1, sigma);
                        }
                        y[a][k][0] = cadd(y[a][k][0], z0);
                        y[a][k][1] = cadd(y[a][k][1], z1);
                    }
                }
            }
            __syncthreads();   // every read of the transmit samples is done: overwrite in place
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                const int m0 = 2 * (tid_c + kPipeBlock * k);
                if (m0 < N) {
                    const int q0 = lds_swz<true>(m0);
                    const int q1 = lds_swz<true>(m0 + 1);
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        s_x[a * N + q0] = y[a][k][0];
                        s_x[a * N + q1] = y[a][k][1];
                    }
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dif_r<T, N, false, kPipeBlock, true, true>(s_x, NB, N, twr);
            else
                fft_dif<T, N, false, kPipeBlock, true, true>(s_x, NB, N, s_tw);   // bins, digit-reversed positions
            // ---- receive: one-tap equaliser from the tap means, demodulate, count -- one subcarrier per thread ----
            const int tid_r = opaque(tid0);
            for (int d = tid_r; d < U; d += kPipeBlock) {
                const int f = ofdm_bin(d, N, U);
                cx<T> h[NB];
#pragma unroll
                for (int a = 0; a < NB; ++a) h[a] = mk<T>(0, 0);
                for (int s = 0; s < S; ++s) {
                    const cx<T> w = s_tw[(f * pp.tap_delay[s]) & (N - 1)];
#pragma unroll
                    for (int a = 0; a < NB; ++a) h[a] = cfma(s_mean[a * S + s], w, h[a]);
                }
                const int bin = lds_swz<true>(fft_pos_of_index<N>(f));
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const cx<T> eq = cdivide(cscale(s_x[a * N + bin], rx_scale), h[a]);
                    const unsigned x = (unsigned)((int)s_idx[a * U + d] ^ demod_one(mp, s_table, s_grid, eq));
                    se[a] += (x != 0u);
                    be[a] += __popc(x);
                }
            }
        }
        // ---- per-slot totals: wave sums, then one thread folds the waves ----
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            const unsigned s1 = wave_sum_u32(se[a]), b1 = wave_sum_u32(be[a]);
            if ((tid0 & 63) == 0) {
                s_red[((tid0 >> 6) * NB + a) * 2] = s1;
                s_red[((tid0 >> 6) * NB + a) * 2 + 1] = b1;
            }
        }
        __syncthreads();
        if (tid0 == 0) {
            for (int a = 0; a < NB; ++a) {
                if (base + a >= count) break;
                unsigned st = 0, bt = 0;
                for (int w = 0; w < kPipeBlock / 64; ++w) {
                    st += s_red[(w * NB + a) * 2];
                    bt += s_red[(w * NB + a) * 2 + 1];
                }
                wg_account(totals, st, bt, false, base + a, sym_out, bit_out);
            }
        }
    }
    if (tid0 == 0)
        wg_flush(totals, counters, (unsigned long long)U * pp.n_ofdm_sym,
                 (unsigned long long)U * pp.n_ofdm_sym * mp.bits);
}

template <typename T, int N>
int run_siso_tdl_batch_impl(mcle_ctx* ctx, SisoTdlParams pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                            mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int NB = 4;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    const size_t PS = (size_t)pp.n_taps * NB;
    const size_t ray_elems = (PS * pp.L * 3 + 1) / 2;            // {re, im, theta} per ray, in complex elements
    pp.x_elems = (int)(ray_elems > (size_t)NB * N ? ray_elems : (size_t)NB * N);
    const size_t lds = (size_t)(pp.x_elems + N + PS * (pp.K + 1) + PS + 2 * NB * (pp.dmax > 0 ? pp.dmax : 1) + kMaxTable) *
                           sizeof(cx<T>) +
                       2 * NB * (kPipeBlock / 64) * sizeof(unsigned) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) + (size_t)NB * pp.num_used + 16;
    if (lds > 160 * 1024) return MCLE_E_UNSUPPORTED;
    auto kern = k_run_ofdm_tdl_batch<T, N, NB>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    const uint64_t cap = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t passes = (count + NB - 1) / NB;
    const unsigned grid = (unsigned)(passes < cap ? passes : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first, count,
                       (const cx<T>*)tw, d_counters, d_sym, d_bit);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

// Returns MCLE_E_UNSUPPORTED (nothing launched) when the configuration is outside the batched kernel's envelope
// (FFT size, Doppler, LDS): the caller (pipelines.hip: mcle_run_ofdm_tdl) then runs the single-realization kernel.
int run_ofdm_tdl_batched(mcle_ctx* ctx, int dtype, const mcle_ofdm_tdl_cfg* cfg, uint64_t seed, uint64_t first,
                         uint64_t count, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    SisoTdlParams pp;
    pp.cp = cfg->cp_size;
    pp.num_used = cfg->num_used;
    pp.n_ofdm_sym = cfg->n_ofdm_sym;
    pp.n_taps = cfg->n_taps;
    pp.L = cfg->L;
    pp.noise_var = cfg->noise_var;
    pp.Fd = cfg->Fd;
    pp.Ts = cfg->Ts;
    {   // numpy.arange(t0, ..., Ts*1.0000000001): delta = fl(fl(t0 + step) - t0), t0 = Ts (fading_generators.py:459-462)
        volatile double step = cfg->Ts * 1.0000000001;
        volatile double nxt = cfg->Ts + step;
        pp.dt = nxt - cfg->Ts;
    }
    pp.dmax = 0;
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) {
        pp.tap_amp[i] = i < cfg->n_taps ? std::sqrt(cfg->tap_power[i]) * std::sqrt(1.0 / (double)cfg->L) : 0.0;
        pp.tap_delay[i] = i < cfg->n_taps ? cfg->tap_delay[i] : 0;
        if (i < cfg->n_taps) {
            if (cfg->tap_delay[i] < 0 || cfg->tap_delay[i] >= cfg->fft_size) return MCLE_E_UNSUPPORTED;
            if (cfg->tap_delay[i] > pp.dmax) pp.dmax = cfg->tap_delay[i];
        }
    }
    const int W = cfg->fft_size + cfg->cp_size;
    const double xc = 0.5 * (double)(W - 1);
    const double z = 2.0 * 3.14159265358979323846 * cfg->Fd * pp.dt * (xc + (double)pp.dmax);
    const double tol = dtype == MCLE_F32 ? 1e-8 : 1e-17;
    int K = 1;
    double term = z * z / 2.0;
    while (term > tol && K < kSisoMaxOrder + 1) {
        ++K;
        term *= z / (double)(K + 1);
    }
    if (K > kSisoMaxOrder) return MCLE_E_UNSUPPORTED;
    if (K < 2) K = 2;
    pp.K = K;
    {
        long double acc[kSisoMaxOrder + 1] = {0.0L};
        for (int j = 0; j < W; ++j) {
            const long double x = (long double)j - (long double)xc;
            long double xp = 1.0L;
            for (int m = 0; m <= K; ++m) {
                acc[m] += xp;
                xp *= x;
            }
        }
        for (int m = 0; m <= kSisoMaxOrder; ++m) pp.mom[m] = m <= K ? (double)(acc[m] / (long double)W) : 0.0;
    }
#define MCLE_RUN(N_)                                                                                                   \
    if (cfg->fft_size == N_)                                                                                           \
        return dtype == MCLE_F32 ? run_siso_tdl_batch_impl<float, N_>(ctx, pp, cfg->demod_method, seed, first, count,  \
                                                                      d_counters, d_sym, d_bit)                        \
                                 : run_siso_tdl_batch_impl<double, N_>(ctx, pp, cfg->demod_method, seed, first, count, \
                                                                       d_counters, d_sym, d_bit);
    MCLE_RUN(64) MCLE_RUN(128) MCLE_RUN(256) MCLE_RUN(512) MCLE_RUN(1024) MCLE_RUN(2048)
#undef MCLE_RUN
    return MCLE_E_UNSUPPORTED;
}

}  // namespace mcle
