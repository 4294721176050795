// siso_tdl_hw.hpp -- config 3 at fft_size 2048 with TWO WAVEFRONTS PER REALIZATION (k_run_ofdm_tdl_hw, round 6) and its launcher.
//
// The one-realization-per-wavefront kernel (siso_tdl_wave.hpp) holds 32 samples per lane at 2048 points and runs its transforms as
// six radix-4 / radix-2 stages through 33 KiB (complex128) of planes per wavefront: the LDS admits four wavefronts per CU -- ONE per
// SIMD -- and the family fell to 0.54 (complex128) / 0.66 (complex64) of the 1024 kernel's per-subcarrier rate.
// Here wavefront j in {0, 1} of a pair owns the time samples n = 2 m + j (the part-wave decomposition of config 4's family,
// pipeline_mimo_pw.hip, NW = 2):
//   * transmit: x[2 m + j] = sum_k' (X[k'] + (-1)^j X[k' + 1024]) w2048^(-j k') w1024^(-m k') -- the first radix-2 stage is formed
//     per wavefront straight from the label bytes (two table look-ups per element), then the 1024-point radix-16 register passes of
//     fft_r16.hpp, results in registers;
//   * the time signal as two PARITY PLANES behind their prefixes: plane q holds x[2 m + q] at Pp + m, so the delayed sample
//     x[n - d] of n = 2 m + j is plane (j - d) & 1 at m + floor((j - d) / 2) -- a wave-uniform plane and offset per tap, consecutive
//     lanes on consecutive words, exactly the reads of the 1024 kernel; a wavefront writes only its own plane;
//   * noise: NOISE block p holds the samples (2 p, 2 p + 1) = the SAME lane and register of the two wavefronts; each draws half of
//     the blocks and hands the partner its two words through the (by then dead) planes -- every block computed once, the ledger
//     unchanged; an odd row start takes unpaired draws as in the other kernels;
//   * receive: each wavefront transforms its own 1024 samples (radix-16 passes, inputs from registers), then the last radix-2
//     stage Y[k' + 1024 q] = A0[k'] + (-1)^q w2048^k' A1[k'] is the one exchange: wavefront q reads both partial transforms at its
//     positions and equalises / decides / counts the 1024 bins of its half of the band.
// The same LDS per realization as before (the two parity planes are the old planes), twice the wavefronts: a workgroup is two
// realizations = four wavefronts, two (complex128) / three (complex64) workgroups per CU, and the transforms are the 1024 kernel's.
// Six workgroup barriers per symbol (signal, channel reads done, noise words, the first receive pass between its
// arithmetic and its stores, partial transforms, next symbol), four wavefronts each, two or three workgroups per CU to cover them.
// Arithmetic outside the transforms as in k_run_ofdm_tdl_wave (Horner per tap in tap order, the equaliser's quotient, decisions);
// tests/test_gpu_tdl_wave.py holds the complex128 counts to the oracle's and to that kernel's per realization.
#pragma once
#include "siso_tdl_wave.hpp"

namespace mcle {

// (timing bound, experiment builds only: -DMCLE_HW_NOBAR compiles the symbol loop WITHOUT its workgroup barriers -- wrong results)
#if defined(MCLE_EXPERIMENTS) && defined(MCLE_HW_NOBAR)
#define MCLE_HW_BARRIER() r16_wave_sync()
#define MCLE_HW_BAR false
#else
#define MCLE_HW_BARRIER() __syncthreads()
#define MCLE_HW_BAR true
#endif

template <typename T, int KT, int WPS, int NRW>
__global__ __launch_bounds__(128 * NRW, WPS) void k_run_ofdm_tdl_hw(SisoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first, uint64_t count,
                                                              const cx<T>* __restrict__ g_twN, const cx<T>* __restrict__ g_twH,
                                                              const cx<T>* __restrict__ g_polys, mcle_counters* counters,
                                                              uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    constexpr int N = 2048, H = 1024, R = 16, TB = 128 * NRW;                // NRW: realizations per workgroup
    auto swz = [](int e) { return lds_swz16f(e); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rz = w >> 1, j = w & 1;                                        // realization slot of the workgroup, time parity
    const int S = pp.n_taps, K = KT;
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    // a wavefront's memory: two planes of `pitch` = H + Pp scalars (Pp >= half the largest tap delay).  During the transforms a
    // plane holds the H swizzled elements of the wavefront's partial transform; between them it is parity plane j of the time signal
    // behind its prefix: x[2 m + j] at Pp + m, x[2 (m + H) + j] once more at Pp + m for -Pp <= m < 0.
    const int pitch = pp.x_elems, Pp = pitch - H;
    T* s_all = reinterpret_cast<T*>(smem);                                   // [NRW][2 wavefronts][2][pitch]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_all + NRW * 4 * pitch);     // [M rounded to 2]
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_table + ((mp.M + 1) & ~1));   // [G * G]
    cx<T>* s_twk = reinterpret_cast<cx<T>*>(s_grid + ((mp.grid.G * mp.grid.G + 1) & ~1));   // [R][kWaveMaxTaps] w2048^(F(64 k) d_s)
    cx<T>* s_w2 = s_twk + R * kWaveMaxTaps;                                  // [R] w2048^(F(64 k)): the last stage's twiddle, uniform part
    unsigned char* s_idx_all = reinterpret_cast<unsigned char*>(s_w2 + R);   // [NRW][U rounded to 16]
    const int idx_pitch = (U + 15) & ~15;
    unsigned* s_part = reinterpret_cast<unsigned*>(s_idx_all + NRW * idx_pitch);   // [2 NRW wavefronts][2]
    T* rbase = s_all + rz * 4 * pitch;                                       // this realization's four planes
    T* pr = rbase + j * 2 * pitch;                                           // my transform planes: re [0, H), im [H, 2 H)
    T* pi = pr + H;
    T* xr = pr;                                                              // my parity plane: re [0, pitch), im [pitch, 2 pitch)
    T* xi = pr + pitch;
    unsigned char* s_idx = s_idx_all + rz * idx_pitch;
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, TB);
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    __shared__ WgTotals totals[NRW];
    if (lane == 0 && j == 0) wg_zero(totals[rz]);
    for (int i = (int)threadIdx.x; i < R * kWaveMaxTaps; i += TB) {
        const int k = i / kWaveMaxTaps, ts = i % kWaveMaxTaps;
        const int fk = fft_index_of_pos<H>(64 * k);
        cx<T> v = mk<T>(0, 0);
#pragma unroll
        for (int q = 0; q < kWaveMaxTaps; ++q)
            if (q == ts && q < S) v = g_twN[(fk * pp.tap_delay[q]) & (N - 1)];
        s_twk[i] = v;
        if (ts == 0) s_w2[k] = g_twN[fk];
    }
    __syncthreads();

    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const double xc = 0.5 * (double)(W - 1);
    const int n_coef = S * (K + 1), rec_len = n_coef + S;
    // the register passes' lane twiddles: complex128 fetches them ahead of each transform (48 registers it does not have across the
    // channel and the noise); complex64 keeps them (24 registers)
#ifndef MCLE_HW_F32_TWFETCH
#define MCLE_HW_F32_TWFETCH 1
#endif
    constexpr bool TWFETCH = sizeof(T) == 8 || MCLE_HW_F32_TWFETCH;   // (complex64 with the twiddles resident, 8 spilled registers: 5.85 against 5.87e7)
    R16Tw64<T> tw16;
    if constexpr (!TWFETCH) tw16 = load_r16_tw<T>(g_twH, lane);
    const cx<T> wl = g_twN[fft_index_of_pos<H>(lane)];                       // w2048^(F(lane)): the last stage's twiddle, lane part
    int dly[kWaveMaxTaps];
#pragma unroll
    for (int s = 0; s < kWaveMaxTaps; ++s) dly[s] = s < S ? pp.tap_delay[s] : 0;
    const int hU = U / 2;
    // data position of bin f (inverse of ofdm_bin), -1 where the bin is not used
    auto data_of_bin = [&](int f) -> int {
        if (U == N) return (f + N / 2) & (N - 1);
        const bool neg = f >= N - hU, pos = f >= 1 && f <= hU;
        return neg ? f - (N - hU) : (pos ? hU + f - 1 : -1);
    };

    for (uint64_t r0 = (uint64_t)blockIdx.x * NRW; r0 < count; r0 += (uint64_t)gridDim.x * NRW) {
        const bool live = r0 + rz < count;                                   // (an odd count: the last slot repeats a realization, unaccounted)
        const uint64_t rl = live ? r0 + rz : count - 1;
        const Rng rng(seed, first + rl);
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            int gi = opaque(lane);
            const cx<T>* __restrict__ g_rec = g_polys + (rl * pp.n_ofdm_sym + os) * (uint64_t)rec_len;
            const cx<T> myrec = gi < rec_len ? g_rec[gi] : mk<T>(0, 0);
            const cx<T> myrec2 = gi + 64 < rec_len ? g_rec[gi + 64] : mk<T>(0, 0);
            auto rec_at = [&](int q) -> cx<T> {
                return q < 64 ? mk<T>(lane_value(myrec.x, q), lane_value(myrec.y, q))
                              : mk<T>(lane_value(myrec2.x, q - 64), lane_value(myrec2.y, q - 64));
            };
            MCLE_HW_BARRIER();                                                // B0: the previous symbol's equaliser has read planes and labels
            // ---- labels: EACH wavefront of the pair draws all of the symbol's DATA blocks (two per lane) and stores all labels -- the
            //      partner stores the same bytes, a wavefront's own LDS traffic is in order, and a barrier is saved ----
            const uint64_t n_first = (uint64_t)os * U, n_last = n_first + U;
            for (uint64_t blk = (n_first >> 4) + (uint64_t)gi; blk <= ((n_last - 1) >> 4); blk += 64) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if ((blk << 4) >= n_first && (blk << 4) + 16 <= n_last && ((n_first & 15) == 0)) {
                    *reinterpret_cast<uint4*>(s_idx + (int)((blk << 4) - n_first)) =
                        make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u), dw.w[2] & (mask * 0x01010101u),
                                   dw.w[3] & (mask * 0x01010101u));
                    continue;
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint64_t n = (blk << 4) + b;
                    if (n >= n_first && n < n_last) s_idx[(int)(n - n_first)] = (unsigned char)((dw.w[b >> 2] >> ((b & 3) * 8)) & mask);
                }
            }
            if constexpr (TWFETCH) tw16 = load_r16_tw<T>(g_twH, opaque(lane));   // in flight behind the look-ups
            r16_wave_sync();                                                // my stores of the labels before my loads
            // ---- transmit: A_j[k'] = (X[k'] + (-1)^j X[k' + H]) w^(-j k') at position p (k' = F(p)) of my planes ----
            gi = opaque(lane);
            int f_lane = fft_index_of_pos<H>(gi);                            // bin (mod H) of position gi (from the opaque lane: what
#pragma unroll                                                              // follows from it is NOT hoisted out of the loops into registers)
            for (int c = 0; c < R; ++c) {
                const int kq = f_lane | fft_index_of_pos<H>(64 * c);
                const int d0 = data_of_bin(kq), d1 = data_of_bin(kq + H);
                const cx<T> X0 = d0 >= 0 ? s_table[s_idx[d0]] : mk<T>(0, 0);
                const cx<T> X1 = d1 >= 0 ? s_table[s_idx[d1]] : mk<T>(0, 0);
                cx<T> a = j ? mk<T>(X0.x - X1.x, X0.y - X1.y) : mk<T>(X0.x + X1.x, X0.y + X1.y);
                if (j) {
                    const cx<T> tw = cmul(wl, s_w2[c]);                      // w^(k'), forward; the inverse transform takes its conjugate
                    a = mk<T>(a.x * tw.x + a.y * tw.y, a.y * tw.x - a.x * tw.y);
                }
                const int sl = swz(gi) ^ swz(64 * c);
                pr[sl] = a.x * tx_scale;
                pi[sl] = a.y * tx_scale;
            }
            r16_wave_sync();
            cx<T> y[R];                                                     // element m = gi + 64 c (time sample 2 m + j) in y[c]
            r16_dit<T, true, true, false, true>(pr, pi, lane, tw16, g_twH, y);
            r16_wave_sync();                                                // every lane's reads of the planes are issued
            gi = opaque(lane);
#pragma unroll
            for (int c = 0; c < R; ++c) {                                   // -> parity plane j behind its prefix
                xr[Pp + gi + 64 * c] = y[c].x;
                xi[Pp + gi + 64 * c] = y[c].y;
            }
#pragma unroll
            for (int c = R - 4; c < R; ++c)                                 // the prefix: the last Pp samples once more (Pp <= 256)
                if (gi + 64 * c >= H - Pp) {
                    xr[gi + 64 * c - (H - Pp)] = y[c].x;
                    xi[gi + 64 * c - (H - Pp)] = y[c].y;
                }
            MCLE_HW_BARRIER();                                                // B2: both parity planes are in place
            // ---- channel: y[n] = sum_s g_s(q) x[n - d_s], n = 2 m + j, q = cp + n - d_s ----
#pragma unroll
            for (int c = 0; c < R; ++c) y[c] = mk<T>(0, 0);
#pragma unroll
            for (int s = 0; s < kWaveMaxTaps; ++s) {
                if (s >= S) break;
                const int d = dly[s];
                cx<T> cc[KT + 1];
#pragma unroll
                for (int m = 0; m <= KT; ++m) cc[m] = rec_at(s * (K + 1) + m);
                const int e = j - d, q = e & 1, o = e >> 1;                 // x[2 m + j - d] = plane q at m + o (o = floor(e / 2) >= -Pp)
                const T* xdr = rbase + q * 2 * pitch + (Pp + gi + o);
                const T* xdi = xdr + pitch;
                const T x0 = sizeof(T) == 8 ? (T)((double)(cp + 2 * gi + e) - xc) : (T)(cp + 2 * gi + e) - (T)xc;   // exact (half-)integers
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const cx<T> xv = mk<T>(xdr[64 * c], xdi[64 * c]);
                    chan_step<KT>(y[c], cc, x0 + (T)(128 * c), xv);
                }
            }
            // ---- noise: sample sym0 + cp + n of the NOISE stream ----
            const uint64_t nbase = sym0 + (uint64_t)cp;
            gi = opaque(lane);
            if ((nbase & 1) == 0) {             // block nbase / 2 + m = (my sample m, the partner's sample m): half of the blocks each
                uint2 own[R / 2], peer[R / 2];
#pragma unroll
                for (int cc = 0; cc < R / 2; ++cc) {
                    const int m = gi + 64 * (8 * j + cc);
                    const Words4 b = rng.block(STREAM_NOISE, (uint32_t)((nbase >> 1) + (uint64_t)m));
                    own[cc] = j ? make_uint2(b.w[2], b.w[3]) : make_uint2(b.w[0], b.w[1]);
                    peer[cc] = j ? make_uint2(b.w[0], b.w[1]) : make_uint2(b.w[2], b.w[3]);
                }
                MCLE_HW_BARRIER();                                            // B3: every read of the parity planes is done
                uint2* mine = reinterpret_cast<uint2*>(pr);                 // [8][64] word pairs for the partner, in my (dead) planes
                const uint2* theirs = reinterpret_cast<const uint2*>(rbase + (j ^ 1) * 2 * pitch);
#pragma unroll
                for (int cc = 0; cc < R / 2; ++cc) mine[cc * 64 + gi] = peer[cc];
                MCLE_HW_BARRIER();                                            // B4: the partner's words for my samples are in its planes
                uint2 got[R / 2];
#pragma unroll
                for (int cc = 0; cc < R / 2; ++cc) got[cc] = theirs[cc * 64 + gi];
                if constexpr (TWFETCH) tw16 = load_r16_tw<T>(g_twH, opaque(lane));   // in flight behind the Box-Muller evaluations
#pragma unroll
                for (int cc = 0; cc < R / 2; ++cc) {                        // registers cc (blocks of wavefront 0) and 8 + cc (of wavefront 1)
                    const uint2 lo = j ? got[cc] : own[cc], hi = j ? own[cc] : got[cc];
                    cx<T> za, zb;
                    if constexpr (sizeof(T) == 8) {
                        za = cn_from_words_lds(lo.x, lo.y, sigma, s_bm);
                        zb = cn_from_words_lds(hi.x, hi.y, sigma, s_bm);
                    } else {
                        za = cn_from_words(lo.x, lo.y, sigma);
                        zb = cn_from_words(hi.x, hi.y, sigma);
                    }
                    y[cc] = cadd(y[cc], za);
                    y[8 + cc] = cadd(y[8 + cc], zb);
                }
            } else {                            // odd row start: a block's samples sit in different registers -- half of every block used
                if constexpr (TWFETCH) tw16 = load_r16_tw<T>(g_twH, opaque(lane));
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const uint64_t i0 = nbase + (uint64_t)(2 * (gi + 64 * c) + j);
                    const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                    const uint32_t x0 = (i0 & 1) ? b.w[2] : b.w[0], x1 = (i0 & 1) ? b.w[3] : b.w[1];
                    cx<T> z;
                    if constexpr (sizeof(T) == 8) z = cn_from_words_lds(x0, x1, sigma, s_bm);
                    else z = cn_from_words(x0, x1, sigma);
                    y[c] = cadd(y[c], z);
                }
            }
            // ---- receive: my 1024 samples -> A_j at digit-reversed positions of my planes.  A workgroup barrier between the first
            //      pass's arithmetic and its stores (r16_pass BAR): the partner has read its words from my planes / every read of the
            //      parity planes is done ----
            r16_dif<T, false, true, false, true, MCLE_HW_BAR>(pr, pi, lane, tw16, g_twH, y);
            MCLE_HW_BARRIER();                                                // B5: both partial transforms are in place
            // ---- last radix-2 stage + one-tap equaliser + decisions: bins f = k' + H j at positions gi + 64 k ----
            gi = opaque(lane);
            f_lane = fft_index_of_pos<H>(gi);
            const int slot_lane = swz(gi);
            const T* a0r = rbase;                                           // wavefront 0's planes
            const T* a1r = rbase + 2 * pitch;                               // wavefront 1's
            cx<T> mean[kWaveMaxTaps];                                       // tap means x w^(F(gi) d_s) x (-1)^(j d_s)
#pragma unroll
            for (int s = 0; s < kWaveMaxTaps; ++s) {
                cx<T> mv = s < S ? cmul(rec_at(n_coef + s), g_twN[(f_lane * dly[s]) & (N - 1)]) : mk<T>(0, 0);
                if (j & dly[s] & 1) mv = mk<T>(-mv.x, -mv.y);
                mean[s] = mv;
            }
            const bool slicer = mp.method == MCLE_DEMOD_QAM_SLICER;
            const bool certpath = !slicer && mp.cert != 0;
            constexpr int GRP = 8;
#pragma unroll
            for (int half = 0; half < R / GRP; ++half) {
                cx<T> eq[GRP];
                int sent[GRP], dec[GRP];
                bool valid[GRP];
#pragma unroll
                for (int b = 0; b < GRP; ++b) {
                    const int k = GRP * half + b;
                    const int f = (f_lane | fft_index_of_pos<H>(64 * k)) + H * j;
                    const int d = data_of_bin(f);
                    valid[b] = d >= 0;
                    const int bin = slot_lane ^ swz(64 * k);
                    const cx<T> a0 = mk<T>(a0r[bin], a0r[H + bin]), a1 = mk<T>(a1r[bin], a1r[H + bin]);
                    const cx<T> t = cmul(cmul(wl, s_w2[k]), a1);
                    eq[b] = j ? mk<T>((a0.x - t.x) * rx_scale, (a0.y - t.y) * rx_scale) : mk<T>((a0.x + t.x) * rx_scale, (a0.y + t.y) * rx_scale);
                    sent[b] = (int)s_idx[d >= 0 ? d : 0];
                }
                cx<T> h[GRP];
#pragma unroll
                for (int b = 0; b < GRP; ++b) h[b] = mk<T>(0, 0);
#pragma unroll
                for (int s = 0; s < kWaveMaxTaps; ++s) {
                    if (s >= S) break;
#pragma unroll
                    for (int b = 0; b < GRP; ++b) h[b] = cfma4(mean[s], s_twk[(GRP * half + b) * kWaveMaxTaps + s], h[b]);
                }
#pragma unroll
                for (int b = 0; b < GRP; ++b) {
                    if constexpr (sizeof(T) == 8) {
                        const T inv = (T)rcp_newton((double)fma(h[b].x, h[b].x, h[b].y * h[b].y));
                        eq[b] = mk<T>(fma(eq[b].x, h[b].x, eq[b].y * h[b].y) * inv, fma(eq[b].y, h[b].x, -(eq[b].x * h[b].y)) * inv);
                    } else {
                        const T inv = __builtin_amdgcn_rcpf(h[b].x * h[b].x + h[b].y * h[b].y);
                        eq[b] = mk<T>((eq[b].x * h[b].x + eq[b].y * h[b].y) * inv, (eq[b].y * h[b].x - eq[b].x * h[b].y) * inv);
                    }
                }
                if (slicer) {
#pragma unroll
                    for (int b = 0; b < GRP; ++b) dec[b] = demod_qam_slicer<T>(eq[b], mp.qam_scale, mp.qam_L, mp.half_bits);
                } else if (certpath) {
                    bool unsure = false;
#pragma unroll
                    for (int b = 0; b < GRP; ++b) {
                        bool sure;
                        dec[b] = demod_cert_any<T>(mp, eq[b], sure);
                        unsure = unsure || (valid[b] && !sure);
                    }
                    if (unsure) {
#pragma unroll
                        for (int b = 0; b < GRP; ++b) dec[b] = demod_one(mp, s_table, s_grid, eq[b]);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < GRP; ++b) dec[b] = demod_one(mp, s_table, s_grid, eq[b]);
                }
#pragma unroll
                for (int b = 0; b < GRP; ++b) {
                    const unsigned x = valid[b] ? (unsigned)(sent[b] ^ dec[b]) : 0u;
                    se += (x != 0u);
                    be += __popc(x);
                }
            }
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        if (lane == 0) {
            s_part[2 * w] = se;
            s_part[2 * w + 1] = be;
        }
        MCLE_HW_BARRIER();                                                    // the pair's two halves of the band
        if (lane == 0 && j == 0 && live)
            wg_account(totals[rz], s_part[2 * w] + s_part[2 * w + 2], s_part[2 * w + 1] + s_part[2 * w + 3], false, rl, sym_out, bit_out);
    }
    wg_flush_waves<NRW>(totals, counters, (unsigned long long)U * pp.n_ofdm_sym, (unsigned long long)U * pp.n_ofdm_sym * mp.bits);
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (the caller goes on to the one-wavefront kernel)
template <typename T, int WPS, int NRW>
int run_siso_tdl_hw(mcle_ctx* ctx, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                    mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int N = 2048, H = 1024, R = 16;
    // every delay inside the prefix (a delay beyond it: the one-wavefront kernel carries the previous symbol's end), orders 2 .. 5
    if (pp.cp < pp.dmax || pp.dmax > 256 || pp.n_taps > kWaveMaxTaps || pp.n_taps * (pp.K + 2) > 128 || pp.K < 2 || pp.K > 5) return MCLE_E_UNSUPPORTED;
    int rc;
    void *twN = nullptr, *twH = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &twN))) return rc;
    if ((rc = ctx->get_twiddles(H, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &twH))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    SisoTdlParams pw = pp;
    pw.x_elems = H + (((pp.dmax + 1) / 2 + 15) & ~15);                       // plane pitch: H + the prefix half the taps' reach
    const size_t lds = (size_t)NRW * 4 * pw.x_elems * sizeof(T) + (((size_t)mp.M + 1) & ~(size_t)1) * sizeof(cx<T>) +
                       (((size_t)mp.grid.G * mp.grid.G + 1) & ~(size_t)1) * sizeof(unsigned long long) + (R * kWaveMaxTaps + R) * sizeof(cx<T>) +
                       NRW * (((size_t)pp.num_used + 15) & ~(size_t)15) + 4 * NRW * sizeof(unsigned) + 16;
    const size_t lds_static = (sizeof(T) == 8 ? (size_t)kBmLdsDoubles * 8 : 8) + NRW * sizeof(WgTotals) + 64;
    int per_cu = (int)((size_t)160 * 1024 / (lds + lds_static));
    if (per_cu * NRW < (sizeof(T) == 8 ? 4 : 6)) return MCLE_E_UNSUPPORTED;  // (a 256-point table in complex128: one workgroup per CU -- no gain)
    if (per_cu > WPS * 2 / NRW) per_cu = WPS * 2 / NRW;                      // WPS wavefronts per SIMD = 2 WPS realizations per CU
    auto kern = k_run_ofdm_tdl_hw<T, 2, WPS, NRW>;
    switch (pp.K) {
        case 2: break;
        case 3: kern = k_run_ofdm_tdl_hw<T, 3, WPS, NRW>; break;
        case 4: kern = k_run_ofdm_tdl_hw<T, 4, WPS, NRW>; break;
        default: kern = k_run_ofdm_tdl_hw<T, 5, WPS, NRW>; break;
    }
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t rec_len = (size_t)pp.n_taps * (pp.K + 2);
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * rec_len;
    uint64_t slice = (2048ull << 20) / (per_real * sizeof(cx<T>));
    slice = slice < NRW ? NRW : (slice / NRW) * NRW;
    if (slice > count) slice = count;
    void* recs = nullptr;
    size_t got = 0;
    const size_t one = (size_t)per_real * sizeof(cx<T>);
    const uint64_t floor_n = slice < 64 * NRW ? slice : 64 * NRW;
    if ((rc = ctx->scratch_upto((size_t)slice * one, (size_t)floor_n * one, &recs, &got))) return rc;
    if (got / one < slice) slice = (got / one / NRW) * NRW;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_tdl_symbol_polys<T>(ctx->stream, pp, N + pp.cp, seed, first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, (n + NRW - 1) / NRW, 4, 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NRW), lds, ctx->stream, pw, mp, seed, first + off, n, (const cx<T>*)twN, (const cx<T>*)twH,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

}  // namespace mcle
