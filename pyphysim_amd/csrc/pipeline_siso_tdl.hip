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
#include <cstdlib>
#include <type_traits>

#include "fft.hpp"
#include "fft16.hpp"
#include "jakes.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "siso_tdl.hpp"
#include "totals.hpp"

namespace mcle {

template <typename T, int N, int NB>
__global__ __launch_bounds__(kPipeBlock, (sizeof(T) == 4 || N <= 1024) ? 3 : 2) void k_run_ofdm_tdl_batch(
    SisoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first, uint64_t count,
    const cx<T>* __restrict__ g_tw, const cx<T>* __restrict__ g_polys, mcle_counters* counters,
    uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = pp.n_taps, K = pp.K, dmax = pp.dmax;
    const int PS = S * NB;                              // fading processes of a pass: slot a, tap s -> a*S + s
    cx<T>* s_x = reinterpret_cast<cx<T>*>(smem);       // [NB][N] (+ slack for the ray scratch of small FFTs)
    // complex64 keeps the twiddle table in LDS; complex128 reads it from global (L1/L2 resident) so that a
    // second and third workgroup fit next to the 16-byte samples
    constexpr bool kTwLds = sizeof(T) == 4;
    cx<T>* s_twbuf = s_x + pp.x_elems;                  // [N] (complex64 only)
    const cx<T>* s_tw = kTwLds ? s_twbuf : g_tw;
    cx<T>* s_coef = s_twbuf + (kTwLds ? N : 0);         // [PS][K+1]
    cx<T>* s_mean = s_coef + PS * (K + 1);              // [PS]
    cx<T>* s_tail = s_mean + PS;                        // [2][NB][dmax] last samples of the previous symbol
    cx<T>* s_table = s_tail + 2 * NB * (dmax > 0 ? dmax : 1);   // [kMaxTable]
    unsigned* s_red = reinterpret_cast<unsigned*>(s_table + kMaxTable);   // [2*NB*4] per-wave partials
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_red + 2 * NB * (kPipeBlock / 64));
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [NB][num_used]

    const int tid0 = threadIdx.x;
    if constexpr (kTwLds)
        for (int k = tid0; k < N; k += kPipeBlock) s_twbuf[k] = g_tw[k];
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];   // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, tid0, kPipeBlock);
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
            // ---- this symbol's tap polynomials and tap means (k_tdl_symbol_polys): slot a's record -> s_coef [a S + s][K + 1],
            //      s_mean [a S + s] (dead since the previous symbol's equaliser; first read after the transmit transform) ----
            {
                const int n_coef = S * (K + 1), rec_len = n_coef + S;
                for (int e = tid; e < NB * rec_len; e += kPipeBlock) {
                    const int a = e / rec_len, r = e - a * rec_len;
                    if (base + a < count) {
                        const cx<T> v = g_polys[((base + a) * pp.n_ofdm_sym + os) * (uint64_t)rec_len + r];
                        if (r < n_coef) s_coef[a * n_coef + r] = v;
                        else s_mean[a * S + (r - n_coef)] = v;
                    }
                }
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
                    if (U == N && (U & 15) == 0) {   // full band on block boundaries: bins bin(d0) ^ j; digit reversal and
                        const int d0 = (int)((blk << 4) - n_first);      // swizzle are XOR-linear: one chain + 16 constants
                        const int p0 = lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(d0, N, U)));
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                            s_idx[a * U + d0 + j] = (unsigned char)tx;
                            s_x[a * N + (p0 ^ lds_swz<true>(fft_pos_of_index<N>(j)))] = cscale(s_table[tx], tx_scale);
                        }
                        continue;
                    }
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
                            cn_pair_lds(rng, STREAM_NOISE, (uint32_t)(i0 >> 1), sigma, z0, z1, s_bm);
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


// ---- the same pass on the matrix cores: f32, N = 1024, every delayed sample inside the symbol's own cyclic prefix ----
// The four slots take the place of the four antennas of pipeline_mimo_mfma.hip (fft16.hpp: 16 x 16 x 4 transform, two
// DFT-16 passes as MFMAs).  What differs is the middle: a tap delay line needs its neighbours' time samples, so P3 parks
// the samples in LDS in a time layout -- in place, each wavefront inside the quarter of the planes it has just read:
// sample m = k + 16 q at k * 64 + (q ^ ((k & 3) << 4)), bank-conflict free for a wavefront's own samples and for
// any common delay -- and the channel reads x[m - d] from there.  Four workgroup barriers per OFDM symbol (after P1,
// after the time samples are parked, after the delayed reads, after P2') instead of fourteen; the rays of the symbol
// are drawn before P1 and folded into tap polynomials by the first wavefront between the first two barriers.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 ld2(const float2* p) {
    const float2 v = *p;
    return f2{v.x, v.y};
}

template <int WAVES, int NB>
__global__ __launch_bounds__(kPipeBlock, WAVES) void k_run_ofdm_tdl_mfma(
    SisoTdlParams pp, ModemParams<float> mp, uint64_t seed, uint64_t first, uint64_t count,
    const float2* __restrict__ g_tw, const float2* __restrict__ g_polys, mcle_counters* counters,
    uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    constexpr int N = kF16N;
    static_assert(NB == 4 || NB == 2, "realization slots per pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = pp.n_taps, K = pp.K;
    const int PS = S * NB;                                             // fading processes of a pass: slot a, tap s -> a*S + s
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    const int tab_len = (mp.M + 15) & ~15;
    float* s_d = reinterpret_cast<float*>(smem);                       // [NB][re plane | im plane]
    float4* s_tab4 = reinterpret_cast<float4*>(s_d + NB * kF16Ant);    // [tab_len] {re, im, |c|^2 / 2, 0} (demodulator)
    float2* s_txtab = reinterpret_cast<float2*>(s_tab4 + tab_len);     // [tab_len] constellation x tx scale
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_txtab + tab_len);     // [NB][U] sent labels
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_idx + ((NB * U + 15) & ~15));
    float2* s_coef = reinterpret_cast<float2*>(s_grid + mp.grid.G * mp.grid.G);     // [PS][K+1]
    float2* s_mean = s_coef + PS * (K + 1);                            // [PS]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_mean + PS);       // [2][4 waves][NB][2]
    constexpr int kPart = 8 * NB;                                      // words of one buffer of partials

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4, gb = g >> 1;
    const float sigma = (float)sqrt(pp.noise_var);
    const float tx_scale = (float)(1.0 / sqrt((double)(U + cp)));
    const float rx_scale = (float)(sqrt((double)(U + cp)) / (double)N);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const uint32_t mask4 = mask * 0x01010101u;
    const double xc = 0.5 * (double)(W - 1);                           // centre of the symbol in local sample units

    for (int m = tid; m < mp.M; m += kPipeBlock) {
        const float2 c = mp.g_table[m];
        s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
        s_txtab[m] = make_float2(c.x * tx_scale, c.y * tx_scale);
    }
    load_grid(mp, s_grid);
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    // ---- per-thread constants (pipeline_mimo_mfma.hip has the same set) ----
    const Dft16Mats mats = dft16_mats(g_tw, lane);
    const int n2 = 16 * w + j;                       // P1 / P1': this lane's column
    const int k1p = 4 * w + (j >> 2), m2p = j & 3;   // P2 / P2': this lane's group (row k1p, residue m2p)
    // middle stage: lane -> butterfly (row k1m, j1m); lanes l and l ^ 1 hold time samples m and m + 1
    const int kkm = ((lane >> 5) << 1) | (lane & 1), j1m = (lane >> 1) & 15, k1m = 4 * w + kkm, par = lane & 1;
    // the three sets of pass twiddles are fetched from the (L1-resident) table at the top of their pass instead of
    // living in 24 registers across the whole loop (1.61 vs 1.63 ms; `opaque` keeps the loads from being hoisted)
    auto load_tw1a = [&](float2 (&tw)[4], int n2_) {
#pragma unroll
        for (int x = 0; x < 4; ++x) tw[x] = g_tw[((4 * g + x) * n2_) & 1023];               // P1 : W1024^{k1 n2}
    };
    auto load_tw2a = [&](float2 (&tw)[4], int m2_) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            tw[x] = g_tw[(16 * (4 * g + x) * m2_) & 1023];                                  // P2 : W64^{j1 m2} ...
            if ((k1p & 1) && (m2_ & 1)) tw[x] = make_float2(-tw[x].x, -tw[x].y);            // ... x the P3 slot order of odd rows
        }
    };
    auto load_tw1b = [&](float2 (&tw)[4], int k1_) {
#pragma unroll
        for (int x = 0; x < 4; ++x) tw[x] = g_tw[((4 * (4 * g + x) + m2p) * k1_) & 1023];   // P2': W1024^{(4 m1 + m2) k1}
    };
    float2 tw1a[4], tw2a[4], tw1b[4], tw2b[3];
#pragma unroll
    for (int m2 = 1; m2 < 4; ++m2) {
        tw2b[m2 - 1] = g_tw[(16 * m2 * j1m) & 1023];                          // P3': W64^{m2 j1} x slot order
        if (par && (m2 & 1)) tw2b[m2 - 1] = make_float2(-tw2b[m2 - 1].x, -tw2b[m2 - 1].y);
    }
    const int plane_g = (g & 1) * kF16Plane;
    const int p1_ld = 64 * gb + (n2 ^ (gb * 36));
    const int p1_st = 256 * g + (n2 ^ (16 * (g & 1)));
    const int p2_base = 64 * k1p + (m2p | f16_swz(k1p));
    const int p2_ld = p2_base ^ (4 * gb);
    const int p2_st = p2_base ^ (16 * g);
    const int mid_off = 64 * k1m + ((4 * j1m) ^ f16_swz(k1m));
    const int m0 = k1m + 16 * j1m;                                            // this lane's time samples: m0 + 256 c
    const int t_sw = (k1m & 3) << 4;
    // symbol scatter, full band: lane (row e = lane >> 2, slot a = lane & 3) of wave w fills bins 64 e + 16 w + (0..15) of
    // slot a from one Philox block -- the wave's OWN 16 columns, so scatter -> P1 and P1' -> next scatter stay wave-local
    // (two slots per pass: rows from lane >> 1, the upper half of the wave has no block to draw)
    constexpr int kSlotBits = NB == 4 ? 2 : 1;
    const int sc_e = lane >> kSlotBits, sc_a = lane & (NB - 1);
    const bool sc_on = sc_e < 16;
    const int sc_d0 = (64 * sc_e + 16 * w + N / 2) & (N - 1);
    const int sc_sw = f16_swz(sc_e);
    const bool full_band = (U == N);

    const uint64_t n_pass = (count + NB - 1) / NB;
    uint64_t it = 0, base_prev = 0;
    __syncthreads();
    for (uint64_t ps = blockIdx.x; ps < n_pass; ps += gridDim.x, ++it) {
        const uint64_t base = ps * NB;                   // slot a carries realization base + a (idle past `count`)
        const int buf = (int)(it & 1);
        unsigned se[NB], be[NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) se[a] = be[a] = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            // ---- this symbol's tap polynomials and tap means (k_tdl_symbol_polys): one value per thread into a register now,
            //      parked in LDS after the first barrier (the previous symbol's equaliser may still be reading s_mean) ----
            float2 poly = make_float2(0.f, 0.f);
            const int rec_len = S * (K + 2);
            if (tid < NB * rec_len) {
                const int a = tid / rec_len, e = tid - a * rec_len;
                if (base + a < count) poly = g_polys[((base + a) * pp.n_ofdm_sym + os) * (uint64_t)rec_len + e];
            }
            // ---- symbols -> bins, stored re<->im swapped (inverse transform by the swap identity) ----
            const uint64_t n_first = (uint64_t)os * U;
            if (full_band) {
              if (sc_on) {
                const Rng rng(seed, first + base + sc_a);
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)((n_first + sc_d0) >> 4));
                const uint32_t wv[4] = {dw.w[0] & mask4, dw.w[1] & mask4, dw.w[2] & mask4, dw.w[3] & mask4};
                *reinterpret_cast<uint4*>(s_idx + sc_a * U + sc_d0) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float2 sym[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) sym[c] = s_txtab[(wv[q] >> (8 * c)) & 0xFFu];
                    const f4 vr = {sym[0].y, sym[1].y, sym[2].y, sym[3].y};
                    const f4 vi = {sym[0].x, sym[1].x, sym[2].x, sym[3].x};
                    const int off = sc_a * kF16Ant + 64 * sc_e + ((16 * w + 4 * q) ^ sc_sw);
                    *reinterpret_cast<f4*>(s_d + off) = vr;
                    *reinterpret_cast<f4*>(s_d + off + kF16Plane) = vi;
                }
              }
                wave_lds_sync();
            } else {           // partial band: zero fill + scatter in block order across the workgroup
                __syncthreads();
                for (int p = tid; p < NB * kF16Ant; p += kPipeBlock) s_d[p] = 0.f;
                __syncthreads();
                const uint64_t n_last = n_first + U;
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const Rng rng(seed, first + base + a);
                    for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += kPipeBlock) {
                        const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
#pragma unroll
                        for (int jj = 0; jj < 16; ++jj) {
                            const uint64_t n = (blk << 4) + jj;
                            if (n >= n_first && n < n_last) {
                                const int tx = (int)((dw.w[jj >> 2] >> ((jj & 3) * 8)) & mask);
                                const int d = (int)(n - n_first);
                                s_idx[a * U + d] = (unsigned char)tx;
                                const float2 c = s_txtab[tx];
                                const int off = a * kF16Ant + f16_pos(ofdm_bin(d, N, U));
                                s_d[off] = c.y;
                                s_d[off + kF16Plane] = c.x;
                            }
                        }
                    }
                }
                __syncthreads();
            }
            // ---- P1: DFT-16 over n1, x W1024^{k1 n2} ----
            load_tw1a(tw1a, opaque(n2));
            dft16_pass<NB>(s_d, plane_g, mats, tw1a, [&](int t) { return (p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t; },
                        [&](int x) { return (p1_st ^ ((x << 2) ^ ((x & 1) << 5))) + 64 * x; }, [&]() {});
            __syncthreads();
            if (tid == 0 && os == 0 && it > 0) {   // every wave is past the previous pass: account it
                const unsigned* q = s_part + (buf ^ 1) * kPart;
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    if (base_prev + a >= count) break;
                    unsigned st = 0, bt = 0;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        st += q[(ww * NB + a) * 2];
                        bt += q[(ww * NB + a) * 2 + 1];
                    }
                    wg_account(totals, st, bt, false, base_prev + a, sym_out, bit_out);
                }
            }
            if (tid < NB * rec_len) {             // record -> s_coef [a*S + s][K + 1], s_mean [a*S + s]
                const int a = tid / rec_len, e = tid - a * rec_len;
                if (e < S * (K + 1)) s_coef[a * S * (K + 1) + e] = poly;
                else s_mean[a * S + (e - S * (K + 1))] = poly;
            }
            // ---- P2: DFT-16 over m1, x W64^{j1 m2} ----
            load_tw2a(tw2a, opaque(m2p));
            dft16_pass<NB>(s_d, plane_g, mats, tw2a, [&](int t) { return p2_ld ^ (8 * t); },
                        [&](int x) { return p2_st ^ (4 * x); }, [&]() {});
            wave_lds_sync();
            // ---- P3 (DFT-4) -> time samples, parked in the time layout inside this wave's own quarter ----
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const f4 R = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + mid_off);
                const f4 I = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + kF16Plane + mid_off);
                const float t0r = R[0] + R[2], t0i = I[0] + I[2], t1r = R[0] - R[2], t1i = I[0] - I[2];
                const float t2r = R[1] + R[3], t2i = I[1] + I[3];
                const float t3r = I[1] - I[3], t3i = R[3] - R[1];     // (z1 - z3) * (-i)
                // planes hold swap(x): true sample = (im plane, re plane); slot s is time index c = s ^ 2 par
                const float xi[4] = {t0r + t2r, t1r + t3r, t0r - t2r, t1r - t3r};
                const float xr[4] = {t0i + t2i, t1i + t3i, t0i - t2i, t1i - t3i};
                __builtin_amdgcn_wave_barrier();                       // every lane of the wave has its operands
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const int c = sl ^ (2 * par);
                    const int off = a * kF16Ant + 64 * k1m + ((j1m + 16 * c) ^ t_sw);
                    s_d[off] = xr[sl];
                    s_d[off + kF16Plane] = xi[sl];
                }
            }
            __syncthreads();
            // ---- channel: y[a][m] = sum_s g[a][s](j) T[a][j], j = cp + m - d_s, + noise ----
            float yr[4][NB], yi[4][NB];            // [slot][realization slot]
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int a = 0; a < NB; ++a) yr[sl][a] = yi[sl][a] = 0.f;
            auto channel = [&](auto k_tag) {      // KT > 0: polynomial order known at compile time
                constexpr int KT = decltype(k_tag)::value;
                for (int s = 0; s < S; ++s) {
                    const int d = pp.tap_delay[s];
                    const int mm = (m0 - d) & (N - 1);
                    const int kk = mm & 15, q0 = mm >> 4, ksw = (kk & 3) << 4;
                    int off[4];
                    f2 xs[4];
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int c = sl ^ (2 * par);
                        off[sl] = 64 * kk + (((q0 + 16 * c) & 63) ^ ksw);
                        const float xv = (float)((double)(cp + m0 + 256 * c - d) - xc);
                        xs[sl] = f2{xv, xv};
                    }
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const float2* c = s_coef + (a * S + s) * (K + 1);
                        const float* pr = s_d + a * kF16Ant;
                        f2 gq[4];                   // tap value at the four samples: Horner on (re, im) pairs (v_pk_fma_f32)
                        if constexpr (KT == 2) {
                            const f2 c0 = ld2(c), c1 = ld2(c + 1), c2 = ld2(c + 2);
#pragma unroll
                            for (int sl = 0; sl < 4; ++sl)
                                gq[sl] = __builtin_elementwise_fma(__builtin_elementwise_fma(c2, xs[sl], c1), xs[sl], c0);
                        } else {
                            const f2 top = ld2(c + K);
#pragma unroll
                            for (int sl = 0; sl < 4; ++sl) gq[sl] = top;
                            for (int m = K - 1; m >= 0; --m) {
                                const f2 cm = ld2(c + m);
#pragma unroll
                                for (int sl = 0; sl < 4; ++sl) gq[sl] = __builtin_elementwise_fma(gq[sl], xs[sl], cm);
                            }
                        }
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl) {
                            const float2 xv = make_float2(pr[off[sl]], pr[off[sl] + kF16Plane]);
                            const float2 acc = cfma(make_float2(gq[sl][0], gq[sl][1]), xv, make_float2(yr[sl][a], yi[sl][a]));
                            yr[sl][a] = acc.x;
                            yi[sl][a] = acc.y;
                        }
                    }
                }
            };
            if (K == 2) channel(std::integral_constant<int, 2>{});
            else channel(std::integral_constant<int, 0>{});
            {   // noise of the samples that survive CP removal: sample sym0 + cp + m of every slot's own NOISE stream;
                // lanes l, l ^ 1 (samples m, m + 1) share a Philox block when that index pair is (even, odd)
                const uint64_t i_base = sym0 + (uint64_t)cp + (uint64_t)m0;
                if (((sym0 + (uint64_t)cp) & 1) == 0) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const Rng rng(seed, first + base + a);
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            const uint64_t i = i_base + 256u * (2 * par + cc);
                            const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i >> 1));
                            const uint32_t k0 = par ? b.w[2] : b.w[0], k1 = par ? b.w[3] : b.w[1];
                            const uint32_t g0 = par ? b.w[0] : b.w[2], g1 = par ? b.w[1] : b.w[3];
                            const float2 keep = cn_from_words(k0, k1, sigma);
                            const float2 give = cn_from_words(g0, g1, sigma);
                            yr[cc][a] += keep.x;
                            yi[cc][a] += keep.y;
                            yr[2 + cc][a] += dpp_swap1(give.x);
                            yi[2 + cc][a] += dpp_swap1(give.y);
                        }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const Rng rng(seed, first + base + a);
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl) {
                            const float2 z = cn_sample<float>(rng, STREAM_NOISE, i_base + 256u * (sl ^ (2 * par)), sigma);
                            yr[sl][a] += z.x;
                            yi[sl][a] += z.y;
                        }
                    }
                }
            }
            __syncthreads();   // every delayed read is done: the planes take the received samples
            // ---- P3': DFT-4 over the slots (their order is folded into tw2b), x W64^{m2 j1} ----
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const float t0r = yr[0][a] + yr[2][a], t0i = yi[0][a] + yi[2][a];
                const float t1r = yr[0][a] - yr[2][a], t1i = yi[0][a] - yi[2][a];
                const float t2r = yr[1][a] + yr[3][a], t2i = yi[1][a] + yi[3][a];
                const float t3r = yi[1][a] - yi[3][a], t3i = yr[3][a] - yr[1][a];
                const float2 v1 = cmul_pk(make_float2(t1r + t3r, t1i + t3i), tw2b[0]);
                const float2 v2 = cmul_pk(make_float2(t0r - t2r, t0i - t2i), tw2b[1]);
                const float2 v3 = cmul_pk(make_float2(t1r - t3r, t1i - t3i), tw2b[2]);
                const f4 vr = {t0r + t2r, v1.x, v2.x, v3.x};
                const f4 vi = {t0i + t2i, v1.y, v2.y, v3.y};
                *reinterpret_cast<f4*>(s_d + a * kF16Ant + mid_off) = vr;
                *reinterpret_cast<f4*>(s_d + a * kF16Ant + kF16Plane + mid_off) = vi;
            }
            wave_lds_sync();
            // ---- P2': DFT-16 over j1, x W1024^{(4 m1 + m2) k1} ----
            load_tw1b(tw1b, opaque(k1p));
            dft16_pass<NB>(s_d, plane_g, mats, tw1b, [&](int t) { return p2_ld ^ (8 * t); },
                        [&](int x) { return p2_st ^ (4 * x); }, [&]() {});
            __syncthreads();
            // ---- P1': DFT-16 over k1 -> bins 64 n1 + n2 (n1 = 4g + x); one-tap equaliser, demodulate, count ----
            {
                float2 yb[4][NB];                 // [x][slot]
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const float* pl = s_d + a * kF16Ant + plane_g;
                    float b[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) b[t] = pl[(p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t];
                    float2 o[4];
                    dft16_mfma(mats, b, o);
#pragma unroll
                    for (int x = 0; x < 4; ++x) yb[x][a] = o[x];
                }
                float2 hq[4][NB];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int a = 0; a < NB; ++a) hq[x][a] = make_float2(0.f, 0.f);
                // W^{f d_s} of tap s + 1 is fetched (L1-resident table) while tap s is accumulated: one table latency in all
                // instead of one per tap (the loop over a run-time tap count is not unrolled)
                float2 wn[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) wn[x] = g_tw[((64 * (4 * g + x) + n2) * pp.tap_delay[0]) & (N - 1)];
                for (int s = 0; s < S; ++s) {
                    float2 wq[4], mean[NB];
#pragma unroll
                    for (int x = 0; x < 4; ++x) wq[x] = wn[x];
                    if (s + 1 < S) {
                        const int dn = pp.tap_delay[s + 1];
#pragma unroll
                        for (int x = 0; x < 4; ++x) wn[x] = g_tw[((64 * (4 * g + x) + n2) * dn) & (N - 1)];
                    }
#pragma unroll
                    for (int a = 0; a < NB; ++a) mean[a] = s_mean[a * S + s];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int a = 0; a < NB; ++a) hq[x][a] = cfma(mean[a], wq[x], hq[x][a]);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int d = ofdm_data_index(64 * (4 * g + x) + n2, N, U);
                    if (d >= 0) {
                        float2 eq[NB];
                        int dec[NB];
#pragma unroll
                        for (int a = 0; a < NB; ++a) {   // y rx_scale / h = y conj(h) (rx_scale / |h|^2), one v_rcp_f32
                            const float2 h = hq[x][a], y = yb[x][a];
                            const float r = rx_scale * __builtin_amdgcn_rcpf(fmaf(h.x, h.x, h.y * h.y));
                            eq[a] = make_float2(fmaf(y.x, h.x, y.y * h.y) * r, fmaf(y.y, h.x, -(y.x * h.y)) * r);
                        }
                        if (mp.method == MCLE_DEMOD_QAM_SLICER) {
#pragma unroll
                            for (int a = 0; a < NB; ++a)
                                dec[a] = demod_qam_slicer<float>(eq[a], mp.qam_scale, mp.qam_L, mp.half_bits);
                        } else if (mp.grid.G > 0 && mp.M > 8) {   // a sweep of <= 8 points beats the cell look-up
                            demod_multi_cert(mp, eq, dec, [&](int (&d_)[NB]) { demod_grid4_multi<NB>(s_tab4, s_grid, mp.grid, mp.M, eq, d_); });
                        } else {
                            demod_multi_cert(mp, eq, dec, [&](int (&d_)[NB]) { demod_mindist_multi<NB>(s_tab4, mp.M, eq, d_); });
                        }
#pragma unroll
                        for (int a = 0; a < NB; ++a) {
                            const unsigned xo = (unsigned)((int)s_idx[a * U + d] ^ dec[a]);
                            se[a] += (xo != 0u);
                            be[a] += __popc(xo);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            const unsigned s1 = wave_sum_u32(se[a]), b1 = wave_sum_u32(be[a]);
            if (lane == 0) {
                s_part[buf * kPart + (w * NB + a) * 2] = s1;
                s_part[buf * kPart + (w * NB + a) * 2 + 1] = b1;
            }
        }
        base_prev = base;
    }
    __syncthreads();
    if (tid == 0) {
        if (it > 0) {
            const unsigned* q = s_part + (int)((it - 1) & 1) * kPart;
            for (int a = 0; a < NB; ++a) {
                if (base_prev + a >= count) break;
                unsigned st = 0, bt = 0;
                for (int ww = 0; ww < 4; ++ww) {
                    st += q[(ww * NB + a) * 2];
                    bt += q[(ww * NB + a) * 2 + 1];
                }
                wg_account(totals, st, bt, false, base_prev + a, sym_out, bit_out);
            }
        }
        wg_flush(totals, counters, (unsigned long long)U * pp.n_ofdm_sym,
                 (unsigned long long)U * pp.n_ofdm_sym * mp.bits);
    }
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (the caller uses k_run_ofdm_tdl_batch)
int run_siso_tdl_mfma(mcle_ctx* ctx, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                      mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    // Register budget and slots per pass (A/B: MCLE_OPT_TDL_MFMA_WAVES): 0 / 2 = two waves per SIMD (256 VGPRs), four
    // realizations per pass; 3 = three waves (168 VGPRs, spills) with four; 32 = three waves with TWO realizations per pass
    const long long wopt = ctx->opt[MCLE_OPT_TDL_MFMA_WAVES];
    const int NB = wopt == 32 ? 2 : 4;
    if (pp.cp < pp.dmax || (pp.num_used & 15) != 0) return MCLE_E_UNSUPPORTED;
    if (ctx->opt[MCLE_OPT_NO_MFMA]) return MCLE_E_UNSUPPORTED;
    const size_t PS = (size_t)pp.n_taps * NB;
    const size_t rec_len = (size_t)pp.n_taps * (pp.K + 2);
    if (NB * rec_len > kPipeBlock || pp.K > kTdlMaxK) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(kF16N, MCLE_F32, &tw))) return rc;
    const ModemParams<float> mp = pipe_modem<float>(ctx, method);
    const size_t tab_len = ((size_t)mp.M + 15) & ~(size_t)15;
    const size_t lds = (size_t)NB * kF16Ant * sizeof(float) + tab_len * (sizeof(float4) + sizeof(float2)) +
                       (((size_t)NB * pp.num_used + 15) & ~(size_t)15) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       (PS * (pp.K + 1) + PS) * sizeof(float2) + 64 * sizeof(unsigned);
    if (lds + 512 > (size_t)160 * 1024 / 2) return MCLE_E_UNSUPPORTED;
    // Round 2, with the f64 ray block inside the kernel: two workgroups per CU at 256 VGPRs (13 spilled) beat three at 168
    // (130 spilled), 1.63 vs 2.04 ms per 131 072 realizations.
    const int waves = (wopt == 3 || wopt == 32) ? 3 : 2;
    auto kern = wopt == 32 ? k_run_ofdm_tdl_mfma<3, 2> : waves == 2 ? k_run_ofdm_tdl_mfma<2, 4> : k_run_ofdm_tdl_mfma<3, 4>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu > waves) per_cu = waves;     // __launch_bounds__(256, WAVES)
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * rec_len;             // complex64 values per realization
    uint64_t slice = (64ull << 20) / (per_real * sizeof(float2));            // <= 64 MiB of records per fading + link pair
    slice = slice < (uint64_t)NB ? (uint64_t)NB : (slice / NB) * NB;
    if (slice > count) slice = (count + NB - 1) / NB * NB;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * per_real * sizeof(float2), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_tdl_symbol_polys<float>(ctx->stream, pp, kF16N + pp.cp, seed, first + off, n, (float2*)recs);
        MCLE_LAUNCH_CHECK();
        const uint64_t passes = (n + NB - 1) / NB;
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, passes);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const float2*)tw, (const float2*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}


// one realization per wavefront (siso_tdl_wave.hpp; pipeline_siso_tdl_wave_f32.hip / _f64.hip): 0 = launched,
// MCLE_E_UNSUPPORTED = outside its envelope (fft_size 256 / 512 / 1024 / 2048, <= 8 taps reaching <= 256 samples back, ...)
int run_siso_tdl_wave_f32(mcle_ctx* ctx, int fft_size, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                          mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
int run_siso_tdl_wave_f64(mcle_ctx* ctx, int fft_size, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                          mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);

template <typename T, int N>
int run_siso_tdl_batch_impl(mcle_ctx* ctx, SisoTdlParams pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                            mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int NB = sizeof(T) == 8 ? 2 : 4;    // complex128: two realizations per pass -> half the LDS, two workgroups per CU
    int rc;
    if constexpr (N == 256 || N == 512 || N == 1024 || N == 2048) {
        // one realization per wavefront: the default where it is the faster kernel (profiles/r04/tdl_family_rates.json: 1024 both
        // arithmetics x1.5, 2048 complex64 x2.7, 256 / 512 complex128 x1.5 / x1.2; the batched complex64 kernels keep 256 / 512, where
        // the per-stage twiddle fetches of the radix-4 wavefront transform are not hidden: x0.54 / x0.86); MCLE_OPT_TDL_KERNEL = 1: the
        // batched kernels everywhere, 2: the wavefront kernel wherever it exists
        // round 5: the wavefront kernel is the faster one at EVERY size in both arithmetics (profiles/r05/tdl_family_rates.json:
        // complex64 256 x1.9, 512 x1.55, 1024 x1.5, 2048 x2.7; complex128 x2.4, x1.6, x1.6, x1.2 -- the complex64 256 / 512 losses
        // of round 4 (x0.63, x0.89) went away with the 64 MiB record slices, which cut a launch of 2^21 short realizations into
        // five pairs of small launches)
        const bool faster = true;
        const long long sel = ctx->opt[MCLE_OPT_TDL_KERNEL];
        if ((sel == 2 || sel == 3 || sel == 4 || (sel == 0 && faster)) && !ctx->opt[MCLE_OPT_NO_MFMA]) {
            rc = sizeof(T) == 8 ? run_siso_tdl_wave_f64(ctx, N, pp, method, seed, first, count, d_counters, d_sym, d_bit)
                                : run_siso_tdl_wave_f32(ctx, N, pp, method, seed, first, count, d_counters, d_sym, d_bit);
            if (rc != MCLE_E_UNSUPPORTED) return rc;
        }
    }
    if constexpr (sizeof(T) == 4 && N == kF16N) {
        rc = run_siso_tdl_mfma(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    const size_t PS = (size_t)pp.n_taps * NB;
    pp.x_elems = NB * N;
    const size_t lds = (size_t)(pp.x_elems + (sizeof(T) == 4 ? N : 0) + PS * (pp.K + 1) + PS + 2 * NB * (pp.dmax > 0 ? pp.dmax : 1) + kMaxTable) *
                           sizeof(cx<T>) +
                       2 * NB * (kPipeBlock / 64) * sizeof(unsigned) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) + (size_t)NB * pp.num_used + 16;
    if (lds > 160 * 1024) return MCLE_E_UNSUPPORTED;
    auto kern = k_run_ofdm_tdl_batch<T, N, NB>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    // two launches per slice of realizations: the symbols' fading records (k_tdl_symbol_polys<T>), then the links
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * pp.n_taps * (pp.K + 2);     // complex values per realization
    uint64_t slice = (64ull << 20) / (per_real * sizeof(cx<T>));                    // <= 64 MiB of records
    slice = slice < (uint64_t)NB ? (uint64_t)NB : (slice / NB) * NB;
    if (slice > count) slice = (count + NB - 1) / NB * NB;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * per_real * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_tdl_symbol_polys<T>(ctx->stream, pp, N + pp.cp, seed, first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        const uint64_t passes = (n + NB - 1) / NB;
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, passes);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const cx<T>*)tw, (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
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
