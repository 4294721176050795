// pipeline_mimo_tdl.hip -- fused pipeline for SURVEY.md section 8(f).1: spatial multiplexing with per-antenna
// OFDM over a frequency-selective, time-varying MIMO TDL channel and one MMSE (or ZF) receive filter per used
// subcarrier.  One workgroup of 256 threads per realization; nothing but integer counters leaves the chip.
//
// Reference path (restated by oracle/chains.py::chain_mimo_ofdm_tdl):
//   TdlMimoChannel / corrupt_data MIMO branch      channels/fading.py:1290-1333, :1107-1117
//   Jakes taps, time axis                          channels/fading_generators.py:421-425, :459-467, :519-522
//   per-symbol mean frequency response             channels/fading.py:513-536 (+ modulators/ofdm.py:545-547)
//   Blast receive filter on every used bin         mimo/mimo.py:577-607
//
// What makes the fusion possible: the staged chain moves the tap tensor g[s][r][a][j] (taps x Nr x Nt x
// samples, 84 % of its HBM bytes) through memory three times.  Here each of the S*Nr*Nt fading processes is
// represented, for one OFDM symbol, by a short polynomial in the sample index,
//     g(x) = amp * sum_l e_l exp(j th_l x) = sum_m c_m x^m,   c_m = amp * sum_l e_l (j th_l)^m / m!,
// x counted from the middle of the symbol, th_l = 2 pi Fd cos(phi_l) dt the per-sample phase advance of ray l
// (~1e-6 rad).  The host picks the order K so that the truncation error (max |th x|)^(K+1)/(K+1)! is below
// the rounding level of the instantiation (K = 3 at config-4 size in f32) and refuses configurations where
// K would exceed kMaxOrder (Doppler phase of > ~0.5 rad across half a symbol) -- callers then use the staged
// operator chain.  The per-symbol tap mean is the same polynomial against precomputed moments of x.
//
// Draw ledger (mcle-philox-v1): DATA symbol n = c*Nt + a; PHASE phi = uniform l*P + p, psi = uniform
// L*P + l*P + p with p = (s*Nr + r)*Nt + a, P = S*Nr*Nt; NOISE sample r*(n + dmax) + j of the faded stream.
#include <type_traits>

#include "fft.hpp"
#include "jakes.hpp"
#include "mimo.hpp"
#include "mimo_tdl.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "qam_pack.hpp"
#include "totals.hpp"

namespace mcle {

// wavefronts per SIMD the register allocation is bounded for = workgroups per CU the LDS admits: three (complex64) / two
// (complex128), but ONE at 2048 x 4 antennas (64 / 128 KiB of samples): bounded for more, those two instantiations spilled
// 113 / 266 registers (928 B of scratch, profiles/r03/kernel_resources.json) for an occupancy they cannot have
template <typename T, int N, int NA> constexpr int mimo_tdl_waves() { return (N >= 2048 && NA >= 4) ? 1 : (sizeof(T) == 4 ? 3 : 2); }
template <typename T, int N, int NA>
__global__ __launch_bounds__(kPipeBlock, (mimo_tdl_waves<T, N, NA>())) void k_run_mimo_ofdm_tdl(
    MimoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first, uint64_t count,
    const cx<T>* __restrict__ g_tw, const cx<T>* __restrict__ g_polys, mcle_counters* counters,
    uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    constexpr int P1 = NA * NA;                 // fading processes per tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = pp.n_taps, K = pp.K, dmax = pp.dmax;
    const int PS = S * P1;                      // fading processes
    cx<T>* s_x = reinterpret_cast<cx<T>*>(smem);       // [NA][N]
    // complex64 keeps the twiddle table in LDS; complex128 reads it from global (L1/L2 resident) so that a
    // second workgroup fits next to the 16-byte samples
    constexpr bool kTwLds = sizeof(T) == 4;
    cx<T>* s_twbuf = s_x + pp.x_elems;                  // [N] (complex64 only)
    const cx<T>* s_tw = kTwLds ? s_twbuf : g_tw;
    cx<T>* s_coef = s_twbuf + (kTwLds ? N : 0);         // [PS][K+1]; complex128: the candidate grid shares it
    const int GG = mp.grid.G * mp.grid.G;
    const int coef_elems = kTwLds ? PS * (K + 1) : max(PS * (K + 1), (GG + 1) / 2);
    cx<T>* s_mean = s_coef + coef_elems;                // [PS]
    cx<T>* s_tail = s_mean + PS;                        // [2][NA][dmax] last samples of the previous symbol
    float4* s_tab4 = reinterpret_cast<float4*>(s_tail + 2 * NA * (dmax > 0 ? dmax : 1));  // [M rounded up to 16]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_tab4);  // f64: the plain table lives in the same place
    const int table_len = (mp.M + 15) & ~15;            // the table region is sized by the constellation
    unsigned* s_red = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(s_tab4) +
                                                  table_len * (sizeof(T) == 4 ? sizeof(float4) : sizeof(cx<T>)));
    // [G*G] candidate grid.  complex128: in the polynomial-coefficient region, which is dead from the end of the
    // channel stage to the top of the next symbol (refilled per symbol from the L2-resident copy)
    unsigned long long* s_grid = kTwLds ? reinterpret_cast<unsigned long long*>(s_red + 16)
                                        : reinterpret_cast<unsigned long long*>(s_coef);
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_red + 16) + (kTwLds ? GG * 8 : 0);   // [NA*num_used]

    const int tid0 = threadIdx.x;
    if constexpr (kTwLds)
        for (int k = tid0; k < N; k += kPipeBlock) s_twbuf[k] = g_tw[k];
    if constexpr (kTwLds) load_grid(mp, s_grid);
    for (int m = tid0; m < mp.M; m += kPipeBlock) {
        const cx<T> c = mp.g_table[m];
        if constexpr (sizeof(T) == 4)
            s_tab4[m] = make_float4((float)c.x, (float)c.y, (float)(0.5 * (c.x * c.x + c.y * c.y)), 0.f);
        else
            s_table[m] = c;
    }
    auto table_at = [&](int m) -> cx<T> {
        if constexpr (sizeof(T) == 4) {
            const float4 c = s_tab4[m];
            return mk<T>(c.x, c.y);
        } else {
            return s_table[m];
        }
    };
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    const int per_sym = U * NA;
    const uint64_t n_total = (uint64_t)pp.n_ofdm_sym * W;        // samples per antenna
    const uint64_t noise_row = n_total + (uint64_t)dmax;          // noise is drawn for the whole faded stream
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NA) / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const T nv_filter = (T)(pp.mmse ? pp.noise_var : 0.0);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const double xc = 0.5 * (double)(W - 1);                      // centre of the symbol in local sample units
    __shared__ WgTotals totals;
    if (tid0 == 0) wg_zero(totals);

    // one butterfly position per thread and stage (N = 4 * threads, radix-4 only): its twiddles live in registers
    constexpr bool kTwRegs = (N == 4 * kPipeBlock) && !FftShape<N>::HAS2 && sizeof(T) == 4;
    cx<T> twr[FftShape<N>::N4][3];
    if constexpr (kTwRegs) {
        __syncthreads();
        fft_twiddle_regs<T, N, kPipeBlock>(s_tw, twr);
    }
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x) {
        const Rng rng(seed, first + rl);
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            const int tid = opaque(tid0);
            __syncthreads();
            // ---- this symbol's tap polynomials and tap means (k_mimo_tdl_symbol_polys): record -> s_coef [PS][K + 1],
            //      s_mean [PS] (both dead since the previous symbol's demodulation; first read after the transmit transform) ----
            {
                const int n_coef = PS * (K + 1), rec_len = n_coef + PS;
                const cx<T>* rec = g_polys + (rl * pp.n_ofdm_sym + os) * (uint64_t)rec_len;
                for (int e = tid; e < rec_len; e += kPipeBlock) {
                    const cx<T> v = rec[e];
                    if (e < n_coef) s_coef[e] = v;
                    else s_mean[e - n_coef] = v;
                }
            }
            // ---- transmit: symbols -> bins (Blast.encode's F-order split + OFDM subcarrier map) ----
            if (U != N) {
                for (int p = tid; p < NA * N; p += kPipeBlock) s_x[p] = mk<T>(0, 0);
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * per_sym;
            const uint64_t n_last = n_first + per_sym;
            // full band, symbol boundaries on DATA blocks: the sixteen symbols of a block are the NA antennas of 16 / NA
            // consecutive subcarriers d0 + t, whose bins are bin(d0) ^ t; digit reversal and swizzle are linear over XOR, so the
            // block needs ONE position chain and sixteen XORs with compile-time constants
            const bool aligned_scatter = U == N && (per_sym & 15) == 0;
            for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += kPipeBlock) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (aligned_scatter) {
                    const int nl0 = (int)((blk << 4) - n_first);
                    const int p0 = lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(nl0 / NA, N, U)));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        s_idx[nl0 + j] = (unsigned char)tx;
                        s_x[(j % NA) * N + (p0 ^ lds_swz<true>(fft_pos_of_index<N>(j / NA)))] = cscale(table_at(tx), tx_scale);
                    }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t n = (blk << 4) + j;
                    if (n >= n_first && n < n_last) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const int nl = (int)(n - n_first);
                        const int a = nl % NA, d = nl / NA;
                        s_idx[nl] = (unsigned char)tx;
                        s_x[a * N + lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(d, N, U)))] = cscale(table_at(tx), tx_scale);
                    }
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dit_r<T, N, true, kPipeBlock, true, FFT_FRESH>(s_x, NA, N, twr);
            else
                fft_dit<T, N, true, kPipeBlock, true, FFT_FRESH>(s_x, NA, N, s_tw);   // bins scattered digit-reversed -> time samples in natural order
            auto time_sample = [&](int a, int i) -> cx<T> {             // IFFT output i of antenna a
                return s_x[a * N + lds_swz<true>(i & (N - 1))];
            };
            const int tid_c = opaque(tid0);      // channel phase: nothing derived from it outlives the next FFT
            // the last dmax samples of this symbol feed the head of the next one
            cx<T>* tail_prev = s_tail + (size_t)(os & 1) * NA * dmax;
            cx<T>* tail_next = s_tail + (size_t)((os + 1) & 1) * NA * dmax;
            if (os + 1 < pp.n_ofdm_sym)
                for (int q = tid_c; q < NA * dmax; q += kPipeBlock) {
                    const int a = q / dmax, i = q - a * dmax;
                    tail_next[q] = time_sample(a, N - dmax + i);
                }
            // ---- channel: y[r][m] = sum_s sum_a g[s][r][a](j) T[a][j],  j = cp + m - d_s (input sample) ----
            constexpr int PAIRS = (N / 2 + kPipeBlock - 1) / kPipeBlock;   // sample pairs per thread
            cx<T> y[NA][PAIRS][2];
#pragma unroll
            for (int r = 0; r < NA; ++r)
#pragma unroll
                for (int k = 0; k < PAIRS; ++k) y[r][k][0] = y[r][k][1] = mk<T>(0, 0);
            // FAST: every input sample lies in this symbol (cp >= max delay) and every thread owns PAIRS full
            // pairs -- no per-sample predicates.  KT > 0: polynomial order known at compile time (Horner unrolled,
            // coefficients fetched together); KT = 0: run-time order.
            auto channel = [&](auto fast_tag, auto k_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                constexpr int KT = decltype(k_tag)::value;
                for (int s = 0; s < S; ++s) {
                    const int d = pp.tap_delay[s];
                    int pos[PAIRS][2];      // >= 0: offset in an antenna's sample row; -1: zero; <= -2: tail slot -2-i
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
#pragma nounroll   // unrolled, the four antennas' loads and Horner chains are hoisted together: > 100 VGPRs spill
                    for (int a = 0; a < NA; ++a) {
                        cx<T> xv[PAIRS][2];
#pragma unroll
                        for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int ps = pos[k][e];
                                if (FAST)
                                    xv[k][e] = s_x[a * N + ps];
                                else
                                    xv[k][e] = ps >= 0 ? s_x[a * N + ps]
                                                       : (ps == -1 ? mk<T>(0, 0) : tail_prev[a * dmax + (-2 - ps)]);
                            }
#pragma unroll
                        for (int r = 0; r < NA; ++r) {
                            const cx<T>* c = s_coef + ((s * NA + r) * NA + a) * (K + 1);
                            cx<T> g[PAIRS][2];
                            if constexpr (KT > 0) {
                                cx<T> cc[KT + 1];
#pragma unroll
                                for (int m = 0; m <= KT; ++m) cc[m] = c[m];
#pragma unroll
                                for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        cx<T> v = cc[KT];
#pragma unroll
                                        for (int m = KT - 1; m >= 0; --m) {
                                            v.x = fma(v.x, xx[k][e], cc[m].x);
                                            v.y = fma(v.y, xx[k][e], cc[m].y);
                                        }
                                        g[k][e] = v;
                                    }
                            } else {
                                const cx<T> top = c[K];
#pragma unroll
                                for (int k = 0; k < PAIRS; ++k) g[k][0] = g[k][1] = top;
                                for (int m = K - 1; m >= 0; --m) {
                                    const cx<T> cm = c[m];
#pragma unroll
                                    for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                                        for (int e = 0; e < 2; ++e) {
                                            g[k][e].x = fma(g[k][e].x, xx[k][e], cm.x);
                                            g[k][e].y = fma(g[k][e].y, xx[k][e], cm.y);
                                        }
                                }
                            }
#pragma unroll
                            for (int k = 0; k < PAIRS; ++k)
#pragma unroll
                                for (int e = 0; e < 2; ++e) y[r][k][e] = cfma(g[k][e], xv[k][e], y[r][k][e]);
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
            // noise of the samples that survive CP removal
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                const int m0 = 2 * (tid_c + kPipeBlock * k);
                if (m0 < N) {
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        const uint64_t i0 = (uint64_t)r * noise_row + sym0 + cp + m0;
                        cx<T> z0, z1;
                        if ((i0 & 1) == 0) {
                            cn_pair<T>(rng, STREAM_NOISE, (uint32_t)(i0 >> 1), sigma, z0, z1);
                        } else {
                            z0 = cn_sample<T>(rng, STREAM_NOISE, i0, sigma);
                            z1 = cn_sample<T>(rng, STREAM_NOISE, i0 + 1, sigma);
                        }
                        y[r][k][0] = cadd(y[r][k][0], z0);
                        y[r][k][1] = cadd(y[r][k][1], z1);
                    }
                }
            }
            __syncthreads();   // every read of the transmit samples is done: overwrite in place
            if constexpr (!kTwLds) load_grid(mp, s_grid);   // ... and of the coefficients (visible after the transform)
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                const int m0 = 2 * (tid_c + kPipeBlock * k);
                if (m0 < N) {
                    const int q0 = lds_swz<true>(m0);
                    const int q1 = lds_swz<true>(m0 + 1);
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        s_x[r * N + q0] = y[r][k][0];
                        s_x[r * N + q1] = y[r][k][1];
                    }
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dif_r<T, N, false, kPipeBlock, true, FFT_FRESH>(s_x, NA, N, twr);
            else
                fft_dif<T, N, false, kPipeBlock, true, FFT_FRESH>(s_x, NA, N, s_tw);   // bins, digit-reversed positions
            // ---- receive: frequency response, filter, decode, demodulate, count -- one subcarrier per thread ----
            const int tid_r = opaque(tid0);
            for (int d = tid_r; d < U; d += kPipeBlock) {
                const int f = ofdm_bin(d, N, U);
                cx<T> H[NA][NA];
#pragma unroll
                for (int r = 0; r < NA; ++r)
#pragma unroll
                    for (int a = 0; a < NA; ++a) H[r][a] = mk<T>(0, 0);
                for (int s = 0; s < S; ++s) {
                    const cx<T> w = s_tw[(f * pp.tap_delay[s]) & (N - 1)];
#pragma unroll
                    for (int r = 0; r < NA; ++r)
#pragma unroll
                        for (int a = 0; a < NA; ++a) H[r][a] = cfma(s_mean[(s * NA + r) * NA + a], w, H[r][a]);
                }
                const int bin = lds_swz<true>(fft_pos_of_index<N>(f));
                cx<T> yb[NA];
#pragma unroll
                for (int r = 0; r < NA; ++r) yb[r] = s_x[r * N + bin];
                cx<T> est[NA];
                int dec[NA];
                const bool ok = blast_solve_t<T, NA, NA>(H, nv_filter, yb, est);   // filter applied, never formed
#pragma unroll
                for (int a = 0; a < NA; ++a) est[a] = ok ? cscale(est[a], rx_scale) : mk<T>(0, 0);   // singular: ZF only
                if constexpr (sizeof(T) == 4 && NA == 4) {
                    if (mp.method == MCLE_DEMOD_QAM_SLICER) {   // the four decisions of the bin packed, compared in the level domain
                        const QamPack qp = qam_pack(mp);
                        const f4q er = {est[0].x, est[1].x, est[2].x, est[3].x}, ei = {est[0].y, est[1].y, est[2].y, est[3].y};
                        const uint32_t sent = *reinterpret_cast<const uint32_t*>(s_idx + d * NA);
                        qam_count4(qam_levels4(er, ei, qp) ^ labels_to_levels(sent, qp), qp, se, be);
                        continue;
                    }
                }
                if (mp.method == MCLE_DEMOD_QAM_SLICER) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) dec[a] = demod_qam_slicer<T>(est[a], mp.qam_scale, mp.qam_L, mp.half_bits);
                } else if constexpr (sizeof(T) == 4) {
                    if (mp.grid.G > 0) {
                        demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_grid4_multi<NA>(s_tab4, s_grid, mp.grid, mp.M, est, d_); });   // the NA streams in lockstep
                    } else {
                        demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_mindist_multi<NA>(s_tab4, mp.M, est, d_); });
                    }
                } else if (mp.grid.G > 0) {
#pragma unroll
                    for (int a = 0; a < NA; ++a) dec[a] = demod_one(mp, s_table, s_grid, est[a]);
                } else {
                    demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_mindist_multi<NA>(s_table, mp.M, est, d_); });
                }
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const unsigned x = (unsigned)((int)s_idx[d * NA + a] ^ dec[a]);
                    se += (x != 0u);
                    be += __popc(x);
                }
            }
        }
        block_sum2(se, be, s_red);
        if (tid0 == 0) wg_account(totals, se, be, false, rl, sym_out, bit_out);
    }
    if (tid0 == 0)
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym,
                 (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
}

template <typename T, int N, int NA>
int run_mimo_tdl_impl(mcle_ctx* ctx, MimoTdlParams pp, int method, uint64_t seed, uint64_t first,
                      uint64_t count, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    const size_t PS = (size_t)pp.n_taps * NA * NA;
    pp.x_elems = NA * N;
    const size_t GG = (size_t)mp.grid.G * mp.grid.G;
    const size_t coef_elems = sizeof(T) == 4 ? PS * (pp.K + 1) : std::max(PS * (pp.K + 1), (GG + 1) / 2);
    const size_t lds = (size_t)(pp.x_elems + (sizeof(T) == 4 ? N : 0) + coef_elems + PS + 2 * NA * (pp.dmax > 0 ? pp.dmax : 1)) * sizeof(cx<T>) +
                       (size_t)((mp.M + 15) & ~15) * (sizeof(T) == 4 ? sizeof(float4) : sizeof(cx<T>)) +
                       16 * sizeof(unsigned) +
                       (sizeof(T) == 4 ? GG * sizeof(unsigned long long) : 0) + (size_t)NA * pp.num_used + 16;
    if (lds > 160 * 1024) {   // the staged operator chain has no such limit
        set_error("configuration needs %zu bytes of LDS (limit 160 KiB)", lds);
        return MCLE_E_UNSUPPORTED;
    }
    auto kern = k_run_mimo_ofdm_tdl<T, N, NA>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    // two launches per slice of realizations: the symbols' fading records (k_mimo_tdl_symbol_polys), then the links; a slice
    // holds <= 256 MiB of records (at 64 MiB the shorter launches cost more than the split gains; 2.5 KiB per realization and symbol in complex64 at 5 taps of 4 x 4, 7.5 KiB in complex128)
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * PS * (pp.K + 2);     // complex values per realization
    uint64_t slice = (256ull << 20) / (per_real * sizeof(cx<T>));
    if (slice < 1) slice = 1;
    if (slice > count) slice = count;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * per_real * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_mimo_tdl_symbol_polys<T, false>(ctx->stream, pp, (int)PS, NA * NA, N + pp.cp, seed, first + off, n, (cx<T>*)recs, 0);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, n);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const cx<T>*)tw, (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// one receive antenna per wavefront (mimo_tdl_wave.hpp; pipeline_mimo_tdl_wave_<arithmetic>_<size>[k].hip): 0 = launched,
// MCLE_E_UNSUPPORTED = outside their envelope
#define MCLE_MIMO_TDL_WAVE_DECL(NAME)                                                                                         \
    int NAME(mcle_ctx* ctx, int nt, int nr, const MimoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count, \
             mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_256) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_512)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_1024) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_1024k)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_2048)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_256) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_512)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_1024) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_1024k)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_2048)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_256k) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_512k) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f32_2048k)
MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_256k) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_512k) MCLE_MIMO_TDL_WAVE_DECL(run_mimo_tdl_wave_f64_2048k)
#undef MCLE_MIMO_TDL_WAVE_DECL
#ifdef MCLE_EXPERIMENTS
int run_mimo_tdl_wave_f32_experiment(int code, mcle_ctx* ctx, int nt, int nr, const MimoTdlParams& pp, int method, uint64_t seed,
                                     uint64_t first, uint64_t count, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
#endif

// the envelope of the wavefront kernels: fft_size 256 .. 2048, <= 8 taps reaching <= 256 samples (and <= N / 2) back -- inside the
// cyclic prefix or beyond it (round 6: the previous symbol's end is carried in LDS; until then a delay beyond the prefix ran the
// cooperative kernel, square channels only)
static bool mimo_tdl_wave_envelope(const MimoTdlParams& pp, int fft_size) {
    return (fft_size == 256 || fft_size == 512 || fft_size == 1024 || fft_size == 2048) && pp.dmax <= 256 &&
           pp.dmax <= fft_size / 2 && pp.n_taps <= 8;
}
// run_time_order: the run-time-order kernels also where the parked-coefficient kernel applies (MCLE_OPT_MIMO_TDL_KERNEL = 2, A/B)
static int run_mimo_tdl_wave(mcle_ctx* ctx, int dtype, int fft_size, int nt, int nr, const MimoTdlParams& pp, bool run_time_order,
                             int method, uint64_t seed, uint64_t first, uint64_t count, mcle_counters* d_counters, uint32_t* d_sym,
                             uint32_t* d_bit) {
    const bool f32 = dtype == MCLE_F32;
#define MCLE_WAVE_CALL(NAME) NAME(ctx, nt, nr, pp, method, seed, first, count, d_counters, d_sym, d_bit)
    // the parked-coefficient kernels (compile-time polynomial order = the benchmark's) first, the run-time-order kernels otherwise
#define MCLE_WAVE_SIZE(N_)                                                                                                  \
    case N_: {                                                                                                              \
        if (!run_time_order) {                                                                                              \
            const int rc = f32 ? MCLE_WAVE_CALL(run_mimo_tdl_wave_f32_##N_##k) : MCLE_WAVE_CALL(run_mimo_tdl_wave_f64_##N_##k); \
            if (rc != MCLE_E_UNSUPPORTED) return rc;                                                                        \
        }                                                                                                                   \
        return f32 ? MCLE_WAVE_CALL(run_mimo_tdl_wave_f32_##N_) : MCLE_WAVE_CALL(run_mimo_tdl_wave_f64_##N_);               \
    }
    switch (fft_size) {
        MCLE_WAVE_SIZE(256) MCLE_WAVE_SIZE(512) MCLE_WAVE_SIZE(1024) MCLE_WAVE_SIZE(2048)
        default: return MCLE_E_UNSUPPORTED;
    }
#undef MCLE_WAVE_SIZE
#undef MCLE_WAVE_CALL
}

}  // namespace mcle

using namespace mcle;

extern "C" int mcle_run_mimo_ofdm_tdl(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_tdl_cfg* cfg, uint64_t seed,
                                      uint64_t first, uint64_t count, mcle_counters* d_counters,
                                      uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    MCLE_REQUIRE(cfg->nt >= 1 && cfg->nt <= cfg->nr && cfg->nr <= 4,
                 "fused MIMO-TDL pipeline: 1 <= Nt <= Nr <= 4 (got %d x %d; use the staged operators otherwise)", cfg->nt, cfg->nr);
    MCLE_REQUIRE(cfg->cp_size >= 0 && cfg->cp_size <= cfg->fft_size,
                 "cp_size must be nonnegative and cannot be greater than fft_size");
    MCLE_REQUIRE(cfg->num_used >= 2 && cfg->num_used % 2 == 0 && cfg->num_used <= cfg->fft_size,
                 "Number of used subcarriers must be a multiple of 2 and at most fft_size");
    MCLE_REQUIRE(cfg->n_ofdm_sym >= 1, "n_ofdm_sym must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "Noise variance must be a non-negative value.");
    MCLE_REQUIRE(cfg->n_taps >= 1 && cfg->n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    MCLE_REQUIRE(cfg->L >= 1 && cfg->L <= 64, "L must be in [1, 64]");
    MCLE_REQUIRE(cfg->Ts > 0.0 && cfg->Fd >= 0.0, "Ts must be positive and Fd non-negative");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    MimoTdlParams pp;
    pp.cp = cfg->cp_size;
    pp.num_used = cfg->num_used;
    pp.n_ofdm_sym = cfg->n_ofdm_sym;
    pp.mmse = cfg->mmse;
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
            MCLE_REQUIRE(cfg->tap_delay[i] >= 0 && cfg->tap_delay[i] < cfg->fft_size,
                         "tap delays must be in [0, fft_size) samples");
            if (cfg->tap_delay[i] > pp.dmax) pp.dmax = cfg->tap_delay[i];
        }
    }
    // polynomial order: truncation (max |theta x|)^(K+1) / (K+1)! below the rounding level of the dtype
    const int W = cfg->fft_size + cfg->cp_size;
    const double xc = 0.5 * (double)(W - 1);
    const double z = 2.0 * 3.14159265358979323846 * cfg->Fd * pp.dt * (xc + (double)pp.dmax);
    const double tol = dtype == MCLE_F32 ? 1e-8 : 1e-17;
    int K = 1;
    double term = z * z / 2.0;          // (K+1 = 2)
    while (term > tol && K < kMaxOrder + 1) {
        ++K;
        term *= z / (double)(K + 1);
    }
    if (K > kMaxOrder) {
        set_error("Doppler phase across half an OFDM symbol is %.3g rad: beyond the fused pipeline's tap model "
                  "(use the staged operator chain)", z);
        return MCLE_E_UNSUPPORTED;
    }
    if (K < 2) K = 2;                   // order 2 is the unrolled fast path
    pp.K = K;
    {
        long double acc[kMaxOrder + 1] = {0.0L};
        for (int j = 0; j < W; ++j) {
            const long double x = (long double)j - (long double)xc;
            long double xp = 1.0L;
            for (int m = 0; m <= K; ++m) {
                acc[m] += xp;
                xp *= x;
            }
        }
        for (int m = 0; m <= kMaxOrder; ++m) pp.mom[m] = m <= K ? (double)(acc[m] / (long double)W) : 0.0;
        for (int s = 0; s < 8; ++s) {           // the wavefront kernels' per-tap moments (mimo_tdl.hpp: mom_tap)
            long double at[kMaxOrder + 1] = {0.0L};
            const long double sh = s < cfg->n_taps ? (long double)pp.tap_delay[s] : 0.0L;
            for (int j = 0; j < W; ++j) {
                const long double x = (long double)j + sh - (long double)xc;
                long double xp = 1.0L;
                for (int m = 0; m <= K; ++m) {
                    at[m] += xp;
                    xp *= x;
                }
            }
            for (int m = 0; m <= kMaxOrder; ++m) pp.mom_tap[s][m] = m <= K ? (double)(at[m] / (long double)W) : 0.0;
        }
        pp.cls_ne = pp.cls_no = 0;
        for (int p = 0; p < 8; ++p) pp.cls_code[p] = -1;
        for (int s = 0; s < cfg->n_taps && s < 8; ++s) {
            const int code = (s << 16) | pp.tap_delay[s];
            if (pp.tap_delay[s] & 1) pp.cls_code[7 - pp.cls_no++] = code;
            else pp.cls_code[pp.cls_ne++] = code;
        }
    }
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    // one receive antenna per wavefront (round 5; every 1 <= Nt <= Nr <= 4 at fft_size 256 .. 2048);
    // MCLE_OPT_MIMO_TDL_KERNEL: 1 = the workgroup-cooperative kernel of rounds 1-4 (Nt = Nr in {2, 4}), 2 = run-time-order kernels
    const long long sel = ctx->opt[MCLE_OPT_MIMO_TDL_KERNEL];
#ifdef MCLE_EXPERIMENTS
    if (sel >= 16 && dtype == MCLE_F32 && cfg->fft_size == 1024 && mimo_tdl_wave_envelope(pp, cfg->fft_size))
        return run_mimo_tdl_wave_f32_experiment((int)sel - 16, ctx, cfg->nt, cfg->nr, pp, cfg->demod_method, seed, first, count, d_counters,
                                                d_sym_err, d_bit_err);
#endif
    if (sel != 1 && mimo_tdl_wave_envelope(pp, cfg->fft_size)) {
        rc = run_mimo_tdl_wave(ctx, dtype, cfg->fft_size, cfg->nt, cfg->nr, pp, sel == 2, cfg->demod_method, seed, first, count,
                               d_counters, d_sym_err, d_bit_err);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    if (!(cfg->nt == cfg->nr && (cfg->nt == 2 || cfg->nt == 4))) {
        set_error("fused MIMO-TDL pipeline: %d x %d at fft_size %d with cp %d / largest delay %d / %d taps is outside the wavefront "
                  "kernels' envelope (fft_size 256 .. 2048, delays <= min(256, fft_size / 2), <= 8 taps) and the cooperative kernel "
                  "takes Nt = Nr in {2, 4} only (use the staged operator chain)",
                  cfg->nt, cfg->nr, cfg->fft_size, cfg->cp_size, pp.dmax, cfg->n_taps);
        return MCLE_E_UNSUPPORTED;
    }
#define MCLE_RUN(N_, NA_)                                                                                       \
    if (cfg->fft_size == N_ && cfg->nt == NA_)                                                                  \
        return dtype == MCLE_F32 ? run_mimo_tdl_impl<float, N_, NA_>(ctx, pp, cfg->demod_method, seed, first,   \
                                                                     count, d_counters, d_sym_err, d_bit_err)   \
                                 : run_mimo_tdl_impl<double, N_, NA_>(ctx, pp, cfg->demod_method, seed, first,  \
                                                                      count, d_counters, d_sym_err, d_bit_err);
    MCLE_RUN(64, 2) MCLE_RUN(64, 4) MCLE_RUN(128, 2) MCLE_RUN(128, 4) MCLE_RUN(256, 2) MCLE_RUN(256, 4)
    MCLE_RUN(512, 2) MCLE_RUN(512, 4) MCLE_RUN(1024, 2) MCLE_RUN(1024, 4) MCLE_RUN(2048, 2) MCLE_RUN(2048, 4)
#undef MCLE_RUN
    set_error("fused MIMO-TDL pipeline supports fft_size in {64, 128, ..., 2048} (got %d)", cfg->fft_size);
    return MCLE_E_INVAL;
}
