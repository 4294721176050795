// siso_tdl_wave.hpp -- config 3 with one realization per wavefront (k_run_ofdm_tdl_wave) and its launcher; compiled per arithmetic
// (pipeline_siso_tdl_wave_f32.hip / _f64.hip: the fifty instantiations were three minutes of one translation unit).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "fft.hpp"
#include "fft_r16.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "pkcx.hpp"
#include "siso_tdl.hpp"
#include "totals.hpp"
#include "wave_lanes.hpp"

namespace mcle {

// ---- config 3, ONE REALIZATION PER WAVEFRONT (round 4, late): every delayed sample behind the symbol's own prefix (round 6: or the previous symbol's end) ----
// The batched kernels above share every transform stage between the 256 threads of a workgroup: a dozen (k_run_ofdm_tdl_batch) or
// four (k_run_ofdm_tdl_mfma) workgroup barriers per OFDM symbol, and the matrix-core kernel -- the default of rounds 2-3 -- left the
// SIMDs idle a third of the time (VALU busy 0.54 + MFMA busy 0.16, profiles/r04/c3_mfma_pmc_summary.json).  Here a wavefront owns a
// realization from its first data word to its error count:
//   * transforms wave-local -- fft_r16.hpp: radix-16 register passes at 1024 (three LDS round trips per transform), radix-4 stages
//     at 256 / 512 / 2048 -- so NOTHING in the loop is a workgroup barrier: the wavefronts of a CU run out of step and fill one
//     another's stalls;
//   * the time signal between the transforms in natural order BEHIND ITS CYCLIC PREFIX, in the memory the swizzled planes
//     occupied (hand-over through registers both ways): x[m - d] is base(tap) + 64 c, immediate offsets, consecutive lanes on
//     consecutive words;
//   * the noise of samples m, m + 1 -- one NOISE block -- belongs to lanes l, l + 1: the even lane draws the blocks of half of the
//     samples a lane holds, the odd lane those of the other half, a DPP lane swap hands over the halves (every block computed
//     once: the draw ledger is unchanged);
//   * nothing wave-uniform is fetched at its point of use: the symbol's record is one coalesced load parked across the lanes and
//     read by v_readlane, tap delays and loop bounds are registers (loops unrolled to kWaveMaxTaps with a uniform guard), the
//     polynomial order is a template parameter;
//   * the equaliser walks POSITIONS, eight at a time, branch-free (loads batched; the certificate's rare "not sure" served once
//     behind the eight); its twiddle w^(f d) = w^(F(lane) d) x w^(F(64 k) d): one gather per lane, tap and symbol + an LDS
//     broadcast, instead of one gather per subcarrier and tap.
// Same arithmetic as k_run_ofdm_tdl_batch operation for operation outside the transforms and the equaliser's twiddle product
// (polynomial Horner, tap order, division), so complex128 counts equal the oracle's like that kernel's (tests/test_gpu_tdl_wave.py).
constexpr int kWaveMaxTaps = 8;
#ifndef MCLE_TDL_WAVE_MAXF
#define MCLE_TDL_WAVE_MAXF 16      // workgroups per resident slot, at most
#endif
// N = 1024: the radix-16 passes with register hand-over on both sides of the channel.  N = 256 / 512 / 2048: radix-4 stages on the
// wavefront's planes (fft_r16.hpp: wave_fft_dif / wave_fft_dit, N / 256 butterfly positions per lane and stage), the same
// hand-over through explicit reads and writes; everything between the transforms is the same code on R = N / 64 samples per lane.
// NWV: wavefronts (= realizations in flight) per workgroup: four, but TWO at 2048 points in complex128, whose 66 KiB of planes per
// wavefront let only two wavefronts share a workgroup's LDS budget (two workgroups per CU: one wavefront per SIMD -- round 5; until
// then that geometry ran the batched kernel only)
template <typename T, int N> constexpr int siso_wave_nwv() { return (N >= 2048 && sizeof(T) == 8) ? 2 : 4; }
template <typename T, int N, int KT, int WPS>
__global__ __launch_bounds__(64 * (siso_wave_nwv<T, N>()), WPS) void k_run_ofdm_tdl_wave(SisoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first,
                                                                uint64_t count, const cx<T>* __restrict__ g_tw,
                                                                const cx<T>* __restrict__ g_polys, mcle_counters* counters,
                                                                uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    constexpr int R = N / 64;                                               // samples (positions, subcarriers) per lane
    constexpr bool R16 = N == 1024;
    constexpr int NWV = siso_wave_nwv<T, N>();
    auto swz = [](int e) { return R16 ? lds_swz16f(e) : lds_swz64(e); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = pp.n_taps, K = KT > 0 ? KT : pp.K;
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    // A wavefront's sample memory: two planes of `pitch` = N + P scalars (P = the largest tap delay rounded up to 16).  During
    // the transforms a plane holds the N swizzled elements (fft_r16.hpp); between them it is the time signal WITH ITS CYCLIC
    // PREFIX in natural order -- xp[P + m] = x[m], xp[j] = x[N - P + j] -- so that x[m - d] is an unswizzled read at a lane-linear
    // address (base per tap + a compile-time offset per sample; conflict free: consecutive lanes, consecutive words).
    const int pitch = pp.x_elems, P = pitch - N;
    T* s_all = reinterpret_cast<T*>(smem);                                   // [NWV wavefronts][2][pitch]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_all + NWV * 2 * pitch);     // [M rounded to 2]   shared, read-only in the loop
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_table + ((mp.M + 1) & ~1));   // [G * G]
    // w^(F(64 k) d_s): the equaliser's twiddle of bin f = F(gi) + F(64 k) is w^(F(gi) d_s) (one gather per lane, tap and symbol)
    // times this wave-uniform factor (an LDS broadcast) -- sixteen table gathers per lane and tap, 64 cache lines each, kept the
    // L1 busier than the SIMDs
    cx<T>* s_twk = reinterpret_cast<cx<T>*>(s_grid + ((mp.grid.G * mp.grid.G + 1) & ~1));   // [R][kWaveMaxTaps]
    unsigned char* s_idx_all = reinterpret_cast<unsigned char*>(s_twk + R * kWaveMaxTaps);        // [4][U rounded to 16]
    const int idx_pitch = (U + 15) & ~15;
    // A tap beyond the cyclic prefix (inter-symbol interference, round 6): positions [0, P - cp) of the signal then hold the END OF
    // THE PREVIOUS SYMBOL -- sample m - d < -cp of the stream is x_prev[N + m - d + cp] -- kept per wavefront in `hist` between the
    // symbols (zeros in front of the first one: the channel's filter starts empty, channels/fading.py:1092-1118)
    const bool isi = pp.dmax > cp;
    const int HL = isi ? pitch - N - cp : 0;
    T* s_hist = reinterpret_cast<T*>(s_idx_all + NWV * idx_pitch) + w * 2 * HL;   // [NWV][2][HL]
    T* pr = s_all + w * 2 * pitch;                                         // transform planes: re [0, N), im [N, 2 N)
    T* pi = pr + N;
    T* xr = pr;                                                             // natural-order signal with prefix: re [0, pitch),
    T* xi = pr + pitch;                                                     // im [pitch, 2 pitch) -- the same memory, never live together
    unsigned char* s_idx = s_idx_all + w * idx_pitch;
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];             // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, 64 * NWV);
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    __shared__ WgTotals totals[NWV];
    if (lane == 0) wg_zero(totals[w]);
    for (int i = (int)threadIdx.x; i < R * kWaveMaxTaps; i += 64 * NWV) {
        const int k = i / kWaveMaxTaps, ts = i % kWaveMaxTaps;
        const int fk = fft_index_of_pos<N>(64 * k);
        cx<T> v = mk<T>(0, 0);
#pragma unroll
        for (int q = 0; q < kWaveMaxTaps; ++q)                              // (static indices into the argument block)
            if (q == ts && q < S) v = g_tw[(fk * pp.tap_delay[q]) & (N - 1)];
        s_twk[i] = v;
    }
    __syncthreads();                                                        // the only workgroup barrier of the kernel

    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const double xc = 0.5 * (double)(W - 1);                                // centre of the symbol in local sample units
    const int n_coef = S * (K + 1), rec_len = n_coef + S;
    [[maybe_unused]] R16Tw64<T> tw16;
    if constexpr (R16) tw16 = load_r16_tw<T>(g_tw, lane);
    // radix-4 sizes in complex64: the lane's stage twiddles in registers (256: 24 registers, 512: 48; 1.11 -> 1.31e8 realizations/s
    // at 256.  complex128 keeps fetching them stage by stage: 48 more registers cost it a wavefront per SIMD, 1.39 -> 1.10e8)
    constexpr int REP4 = N / 256;
    constexpr bool WTW = !R16 && sizeof(T) == 4 && REP4 * FftShape<N>::N4 * 6 <= 48;
    [[maybe_unused]] TwRegs64<T, N> twr4[WTW ? REP4 : 1];
    if constexpr (WTW) {
#pragma unroll
        for (int rep = 0; rep < REP4; ++rep) twr4[rep] = load_tw64<T, N>(g_tw, lane + 64 * rep);
    }
    int dly[kWaveMaxTaps];
#pragma unroll
    for (int s = 0; s < kWaveMaxTaps; ++s) dly[s] = s < S ? pp.tap_delay[s] : 0;

    const uint64_t n_waves = (uint64_t)gridDim.x * NWV;
    for (uint64_t rl = (uint64_t)blockIdx.x * NWV + w; rl < count; rl += n_waves) {
        const Rng rng(seed, first + rl);
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            // the symbol's record (tap polynomials, tap means; rec_len <= 128 values): coalesced loads, value j parked in lane j mod 64,
            // fetched here and first used after the transmit transform -- read at its points of use (scalar loads from the
            // record) its latency stood in front of the channel and the equaliser of every realization
            int gi = opaque(lane);
            const cx<T>* __restrict__ g_rec = g_polys + (rl * pp.n_ofdm_sym + os) * (uint64_t)rec_len;
            const cx<T> myrec = gi < rec_len ? g_rec[gi] : mk<T>(0, 0);
            const cx<T> myrec2 = gi + 64 < rec_len ? g_rec[gi + 64] : mk<T>(0, 0);      // (values 64 .. 127: eight taps of order >= 7)
            auto rec_at = [&](int j) -> cx<T> {
                return j < 64 ? mk<T>(lane_value(myrec.x, j), lane_value(myrec.y, j))
                              : mk<T>(lane_value(myrec2.x, j - 64), lane_value(myrec2.y, j - 64));
            };
            r16_wave_sync();                                                // the previous symbol's equaliser has read the planes
            // ---- transmit: symbols -> bins at digit-reversed positions (the DIT transform takes them from there) ----
            if (U != N) {
                for (int p = gi; p < 2 * N; p += 64) pr[p] = 0;
                r16_wave_sync();
            }
            const uint64_t n_first = (uint64_t)os * U, n_last = n_first + U;
            // 256 / 512 points, full band: the symbol's N / 16 DATA blocks occupy 16 / 32 lanes -- they store the label bytes, and ALL
            // 64 lanes then take N / 64 consecutive labels each for the look-ups and the scatter (round 6, last day: the 16 lanes that
            // drew a block also scattered its sixteen symbols, 128 issue slots at a quarter of the lanes -- 7 % of the 256 kernel)
            bool scattered = false;
            if constexpr (N <= 512) {
                if (U == N) {
                    constexpr int PER = N / 64;                              // labels per lane: 4 or 8
                    if (gi < N / 16) {
                        const Words4 dw = rng.block(STREAM_DATA, (uint32_t)((n_first >> 4) + (uint64_t)gi));
                        *reinterpret_cast<uint4*>(s_idx + 16 * gi) = make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u),
                                                                                dw.w[2] & (mask * 0x01010101u), dw.w[3] & (mask * 0x01010101u));
                    }
                    r16_wave_sync();
                    const int d0 = PER * gi;
                    const int p0 = swz(fft_pos_of_index<N>(ofdm_bin(d0, N, U)));
                    uint32_t lab[PER / 4];
#pragma unroll
                    for (int q = 0; q < PER / 4; ++q) lab[q] = *reinterpret_cast<const uint32_t*>(s_idx + d0 + 4 * q);
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        const int tx = (int)((lab[j >> 2] >> ((j & 3) * 8)) & 0xFFu);
                        const cx<T> c = cscale(s_table[tx], tx_scale);
                        const int pos = p0 ^ swz(fft_pos_of_index<N>(j));   // bin(d0 + j) = bin(d0) ^ j: digit reversal and swizzle are XOR-linear
                        pr[pos] = c.x;
                        pi[pos] = c.y;
                    }
                    scattered = true;
                }
            }
            if (!scattered)
            for (uint64_t blk = (n_first >> 4) + gi; blk <= ((n_last - 1) >> 4); blk += 64) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (U == N && (U & 15) == 0) {           // full band on block boundaries: bin(d0 + j) = bin(d0) ^ j, and digit
                    const int d0 = (int)((blk << 4) - n_first);              // reversal and swizzle are XOR-linear
                    const int p0 = swz(fft_pos_of_index<N>(ofdm_bin(d0, N, U)));
                    *reinterpret_cast<uint4*>(s_idx + d0) = make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u),
                                                                       dw.w[2] & (mask * 0x01010101u), dw.w[3] & (mask * 0x01010101u));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const cx<T> c = cscale(s_table[tx], tx_scale);
                        const int pos = p0 ^ swz(fft_pos_of_index<N>(j));
                        pr[pos] = c.x;
                        pi[pos] = c.y;
                    }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t n = (blk << 4) + j;
                    if (n >= n_first && n < n_last) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const int d = (int)(n - n_first);
                        s_idx[d] = (unsigned char)tx;
                        const cx<T> c = cscale(s_table[tx], tx_scale);
                        const int pos = swz(fft_pos_of_index<N>(ofdm_bin(d, N, U)));
                        pr[pos] = c.x;
                        pi[pos] = c.y;
                    }
                }
            }
            r16_wave_sync();
            cx<T> y[R];                                                     // element gi + 64 c in y[c]
            if constexpr (R16) {
                r16_dit<T, true, true, false, true>(pr, pi, lane, tw16, g_tw, y);   // time samples; the last pass leaves them in registers
            } else {
                wave_fft_dit<T, N, true, WTW>(pr, g_tw, lane, twr4);                   // time samples at swizzled natural positions
                gi = opaque(lane);
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int sl = swz(gi) ^ swz(64 * c);
                    y[c] = mk<T>(pr[sl], pi[sl]);
                }
            }
            r16_wave_sync();                                                // every lane's reads of the planes are issued
            gi = opaque(lane);
#pragma unroll
            for (int c = 0; c < R; ++c) {                                   // -> natural order behind the prefix
                xr[P + gi + 64 * c] = y[c].x;
                xi[P + gi + 64 * c] = y[c].y;
            }
#pragma unroll
            for (int c = (R > 4 ? R - 4 : 0); c < R; ++c)                   // the prefix: the last P samples once more (P <= 256)
                if (gi + 64 * c >= N - P + HL) {
                    xr[gi + 64 * c - (N - P)] = y[c].x;
                    xi[gi + 64 * c - (N - P)] = y[c].y;
                }
            if (isi) {                                                      // in front of the prefix: the previous symbol's end; the
#pragma unroll                                                              // lane that holds sample e now held it then
                for (int c = (R > 4 ? R - 4 : 0); c < R; ++c) {
                    const int j = gi + 64 * c - (N - HL);
                    if (j >= 0) {
                        xr[j] = os > 0 ? s_hist[j] : (T)0;
                        xi[j] = os > 0 ? s_hist[HL + j] : (T)0;
                        s_hist[j] = y[c].x;
                        s_hist[HL + j] = y[c].y;
                    }
                }
            }
            r16_wave_sync();
            // ---- channel: y[m] = sum_s g_s(j) x[j],  j = cp + m - d_s, for this lane's R samples m = gi + 64 c ----
#pragma unroll
            for (int c = 0; c < R; ++c) y[c] = mk<T>(0, 0);
            // (the tap loop is unrolled to kWaveMaxTaps with a wave-uniform guard: the delays and every index are compile-time
            //  register names -- a run-time `pp.tap_delay[s]` is a scalar load from the kernel arguments on the critical path)
#pragma unroll
            for (int s = 0; s < kWaveMaxTaps; ++s) {
                if (s >= S) break;
                const int d = dly[s];
                cx<T> cc[KT > 0 ? KT + 1 : 1];
                if constexpr (KT > 0) {
#pragma unroll
                    for (int m = 0; m <= KT; ++m) cc[m] = rec_at(s * (K + 1) + m);
                }
                const T* xdr = xr + (P + gi - d);                           // x[m - d] = xdr[64 c]: d <= P
                const T* xdi = xi + (P + gi - d);
                // (double) q - xc rounded to T, q = cp + m - d: q and xc are (half-)integers below 2^12 -- exact in float too
                const T x0 = sizeof(T) == 8 ? (T)((double)(cp + gi - d) - xc) : (T)(cp + gi - d) - (T)xc;
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const cx<T> xv = mk<T>(xdr[64 * c], xdi[64 * c]);
                    const T xx = x0 + (T)(64 * c);                          // exact
                    cx<T> g;
                    if constexpr (KT > 0) {                                 // Horner + multiply-add: packed in complex64 (pkcx.hpp)
                        chan_step<KT>(y[c], cc, xx, xv);
                        continue;
                    } else {
                        g = rec_at(s * (K + 1) + K);
                        for (int mm = K - 1; mm >= 0; --mm) {
                            const cx<T> cm = rec_at(s * (K + 1) + mm);
                            g.x = fma(g.x, xx, cm.x);
                            g.y = fma(g.y, xx, cm.y);
                        }
                    }
                    y[c] = cfma4(g, xv, y[c]);
                }
            }
            // ---- noise: sample sym0 + cp + m of the NOISE stream ----
            const uint64_t nbase = sym0 + (uint64_t)cp;
            if ((nbase & 1) == 0) {             // lanes l (even), l + 1 share the block of samples m, m + 1
                const bool odd = (gi & 1) != 0;
#pragma unroll
                for (int j = 0; j < R / 2; ++j) {
                    const int c = j + (odd ? R / 2 : 0);
                    const int m = gi + 64 * c;
                    cx<T> za, zb;
                    cn_pair_lds(rng, STREAM_NOISE, (uint32_t)((nbase + (uint64_t)m) >> 1), sigma, za, zb, s_bm);
                    const T sx = odd ? za.x : zb.x, sy = odd ? za.y : zb.y;      // what the partner needs
                    const T rx = dpp_swap1<T>(sx), ry = dpp_swap1<T>(sy);
                    const cx<T> lo = mk<T>(odd ? rx : za.x, odd ? ry : za.y);    // sample of combination j
                    const cx<T> hi = mk<T>(odd ? zb.x : rx, odd ? zb.y : ry);    // sample of combination R / 2 + j
                    y[j] = cadd(y[j], lo);
                    y[R / 2 + j] = cadd(y[R / 2 + j], hi);
                }
            } else {                            // odd start: a block's samples sit on lanes of different pairs -- half of every block used
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int m = gi + 64 * c;
                    const uint64_t i0 = nbase + (uint64_t)m;
                    const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                    const uint32_t x0 = (i0 & 1) ? b.w[2] : b.w[0], x1 = (i0 & 1) ? b.w[3] : b.w[1];
                    cx<T> z;
                    if constexpr (sizeof(T) == 8) z = cn_from_words_lds(x0, x1, sigma, s_bm);
                    else z = cn_from_words(x0, x1, sigma);
                    y[c] = cadd(y[c], z);
                }
            }
            r16_wave_sync();                                                // every lane's reads of the transmit samples are issued
            if constexpr (R16) {
                // y[q + 4 m'] is element gi + 64 q + 256 m' -- what pass A of the forward transform takes: straight from the registers
                r16_dif<T, false, true, false, true>(pr, pi, lane, tw16, g_tw, y);   // bins at digit-reversed positions
            } else {
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int sl = swz(gi) ^ swz(64 * c);
                    pr[sl] = y[c].x;
                    pi[sl] = y[c].y;
                }
                r16_wave_sync();
                wave_fft_dif<T, N, false, WTW>(pr, g_tw, lane, twr4);
            }
            r16_wave_sync();
            // ---- receive: one-tap equaliser from the tap means, demodulate, count -- POSITIONS gi, gi + 64, ...: position
            //      p = p4 p3 p2 p1 p0 (base 4) holds bin f = p0 p1 p2 p3 p4, so f = F(gi) + F(64 k) with the second term a constant
            //      (the digit reversal is a bit permutation, gi and 64 k share no bits) ----
            gi = opaque(lane);
            const int f_lane = fft_index_of_pos<N>(gi);
            const int slot_lane = swz(gi);
            const int hU = U / 2;
            cx<T> mean[kWaveMaxTaps];                                       // the symbol's tap means (wave-uniform) x w^(F(gi) d_s)
#pragma unroll
            for (int s = 0; s < kWaveMaxTaps; ++s)
                mean[s] = s < S ? cmul(rec_at(n_coef + s), g_tw[(f_lane * dly[s]) & (N - 1)]) : mk<T>(0, 0);
            // Eight subcarriers at a time as ONE straight-line region (loads batched, no branch per subcarrier): an in-order
            // wavefront that stops at every table look-up of every subcarrier spent a third of its time in s_waitcnt.  The
            // certificate's rare "not sure" is collected over the eight and served once, by the table search, behind them.
            const bool slicer = mp.method == MCLE_DEMOD_QAM_SLICER;
            const bool certpath = !slicer && mp.cert != 0;
            constexpr int GRP = R < 8 ? R : 8;
#pragma unroll
            for (int half = 0; half < R / GRP; ++half) {
                cx<T> eq[GRP];
                int sent[GRP], dec[GRP];
                bool valid[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int k = GRP * half + j;
                    const int f = f_lane | fft_index_of_pos<N>(64 * k);
                    int d;                                                  // data position of bin f (inverse of ofdm_bin)
                    if (U == N) {
                        d = (f + N / 2) & (N - 1);
                        valid[j] = true;
                    } else {
                        const bool neg = f >= N - hU, pos = f >= 1 && f <= hU;
                        d = neg ? f - (N - hU) : (pos ? hU + f - 1 : 0);
                        valid[j] = neg || pos;
                    }
                    const int bin = slot_lane ^ swz(64 * k);
                    eq[j] = cscale(mk<T>(pr[bin], pi[bin]), rx_scale);
                    sent[j] = (int)s_idx[d];
                }
                cx<T> h[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) h[j] = mk<T>(0, 0);
#pragma unroll
                for (int s = 0; s < kWaveMaxTaps; ++s) {
                    if (s >= S) break;
#pragma unroll
                    for (int j = 0; j < GRP; ++j) h[j] = cfma4(mean[s], s_twk[(GRP * half + j) * kWaveMaxTaps + s], h[j]);
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    if constexpr (sizeof(T) == 8) {     // one Newton reciprocal (common.hpp) instead of cdivide's two IEEE divisions
                        const T inv = (T)rcp_newton((double)fma(h[j].x, h[j].x, h[j].y * h[j].y));
                        eq[j] = mk<T>(fma(eq[j].x, h[j].x, eq[j].y * h[j].y) * inv, fma(eq[j].y, h[j].x, -(eq[j].x * h[j].y)) * inv);
                    } else {                    // complex64: one reciprocal instead of two divisions
                        const T inv = __builtin_amdgcn_rcpf(h[j].x * h[j].x + h[j].y * h[j].y);
                        eq[j] = mk<T>((eq[j].x * h[j].x + eq[j].y * h[j].y) * inv, (eq[j].y * h[j].x - eq[j].x * h[j].y) * inv);
                    }
                }
                if (slicer) {
#pragma unroll
                    for (int j = 0; j < GRP; ++j) dec[j] = demod_qam_slicer<T>(eq[j], mp.qam_scale, mp.qam_L, mp.half_bits);
                } else if (certpath) {
                    bool unsure = false;
#pragma unroll
                    for (int j = 0; j < GRP; ++j) {
                        bool sure;
                        dec[j] = demod_cert_any<T>(mp, eq[j], sure);
                        unsure = unsure || (valid[j] && !sure);
                    }
                    if (unsure) {
#pragma unroll
                        for (int j = 0; j < GRP; ++j) dec[j] = demod_one(mp, s_table, s_grid, eq[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < GRP; ++j) dec[j] = demod_one(mp, s_table, s_grid, eq[j]);
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const unsigned x = valid[j] ? (unsigned)(sent[j] ^ dec[j]) : 0u;
                    se += (x != 0u);
                    be += __popc(x);
                }
            }
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        if (lane == 0) wg_account(totals[w], se, be, false, rl, sym_out, bit_out);
    }
    wg_flush_waves<NWV>(totals, counters, (unsigned long long)U * pp.n_ofdm_sym, (unsigned long long)U * pp.n_ofdm_sym * mp.bits);   // one flush per workgroup (round 6)
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (the caller goes on to the batched kernels)
template <typename T, int N, int WPS>
int run_siso_tdl_wave_w(mcle_ctx* ctx, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                        mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int R = N / 64;
    if (pp.K > kTdlMaxK) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    // (the prefix copy covers the last four 64-sample blocks; a record is parked in two registers per lane)
    if (pp.dmax > 256 || pp.dmax > N / 2 || pp.n_taps > kWaveMaxTaps || pp.n_taps * (pp.K + 2) > 128) return MCLE_E_UNSUPPORTED;
    SisoTdlParams pw = pp;
    pw.x_elems = N + ((pp.dmax + 15) & ~15);                                // plane pitch: N + the prefix the taps reach into
    constexpr int NWV = siso_wave_nwv<T, N>();
    const size_t lds = (size_t)NWV * 2 * pw.x_elems * sizeof(T) + (((size_t)mp.M + 1) & ~(size_t)1) * sizeof(cx<T>) +
                       (((size_t)mp.grid.G * mp.grid.G + 1) & ~(size_t)1) * sizeof(unsigned long long) + R * kWaveMaxTaps * sizeof(cx<T>) +
                       NWV * (((size_t)pp.num_used + 15) & ~(size_t)15) + 16 +
                       (pp.dmax > pp.cp ? (size_t)NWV * 2 * (pw.x_elems - N - pp.cp) * sizeof(T) : 0);   // the previous symbol's end
    auto kern = k_run_ofdm_tdl_wave<T, N, 2, WPS>;   // the polynomial order is a compile-time constant (2 .. 8: Doppler x symbol
    switch (pp.K) {                                  // length up to ~0.1 turns in complex64; beyond: the batched kernels)
        case 2: kern = k_run_ofdm_tdl_wave<T, N, 2, WPS>; break;
        case 3: kern = k_run_ofdm_tdl_wave<T, N, 3, WPS>; break;
        case 4: kern = k_run_ofdm_tdl_wave<T, N, 4, WPS>; break;
        case 5: kern = k_run_ofdm_tdl_wave<T, N, 5, WPS>; break;
        default:                                     // orders 6 .. 8 at the benchmark size only (build time: 18 kernels less)
            if constexpr (N == 1024) {
                if (pp.K == 6) kern = k_run_ofdm_tdl_wave<T, N, 6, WPS>;
                else if (pp.K == 7) kern = k_run_ofdm_tdl_wave<T, N, 7, WPS>;
                else if (pp.K == 8) kern = k_run_ofdm_tdl_wave<T, N, 8, WPS>;
                else return MCLE_E_UNSUPPORTED;
            } else {
                return MCLE_E_UNSUPPORTED;
            }
    }
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + (sizeof(T) == 8 ? 5 * 1024 : 512)));   // (+ the static Box-Muller tables and totals)
    if (per_cu < 1) return MCLE_E_UNSUPPORTED;
    if (per_cu > WPS * 4 / NWV) per_cu = WPS * 4 / NWV;                      // workgroups of NWV wavefronts, WPS wavefronts per SIMD
    const size_t rec_len = (size_t)pp.n_taps * (pp.K + 2);
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * rec_len;             // complex values per realization
    uint64_t slice = (2048ull << 20) / (per_real * sizeof(cx<T>));           // <= 2 GiB of records per fading + link pair: a bench step
                                                                             // of 2^21 realizations is one dispatch of each kernel
    slice = slice < NWV ? NWV : (slice / NWV) * NWV;
    if (slice > count) slice = count;
    void* recs = nullptr;
    size_t got = 0;
    const size_t one = (size_t)per_real * sizeof(cx<T>);
    const uint64_t floor_n = slice < 64 * NWV ? slice : 64 * NWV;            // scratch_upto: a smaller slice on a crowded device
    if ((rc = ctx->scratch_upto((size_t)slice * one, (size_t)floor_n * one, &recs, &got))) return rc;
    if (got / one < slice) slice = (got / one / NWV) * NWV;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_tdl_symbol_polys<T>(ctx->stream, pp, N + pp.cp, seed, first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        // (round 6: one flush of the counters per WORKGROUP instead of per wavefront -- a workgroup's fixed cost fell, and the grid that
        //  wanted >= 12 passes per workgroup now takes 4: +4 % at 131 072 realizations per launch, profiles/r06/grid_sweep_others.log)
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, (n + NWV - 1) / NWV, 4, MCLE_TDL_WAVE_MAXF);   // (sixteen against eight: +1 % at every size)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWV), lds, ctx->stream, pw, mp, seed, first + off, n, (const cx<T>*)tw,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}
template <typename T, int N>
int run_siso_tdl_wave(mcle_ctx* ctx, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                      mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    // wavefronts per SIMD the registers are bounded for.  FFT 1024: complex128 two (LDS: two workgroups per CU); complex64 three
    // (MCLE_OPT_TDL_KERNEL = 4: four -- the LDS admits a fourth workgroup, but the 128-register bound spills 15 registers:
    //  3.18 against 2.21 ms per 262 144 realizations).  256 / 512: fewer samples per lane, three to six; 2048: complex64 only, two
    //  (132 KiB of planes per workgroup in complex128: the batched kernel serves that one).
    if constexpr (N == 2048) {       // complex128: two wavefronts per workgroup, two workgroups per CU = one wavefront per SIMD
        // (round 6: two wavefronts per REALIZATION -- siso_tdl_hw.hpp, tried first by the callers in pipeline_siso_tdl_wave_*.hip)
        if constexpr (sizeof(T) == 8) return run_siso_tdl_wave_w<T, N, 1>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        else return run_siso_tdl_wave_w<T, N, 2>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
    } else if constexpr (N == 1024) {
        if constexpr (sizeof(T) == 8) return run_siso_tdl_wave_w<T, N, 2>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        else if (ctx->opt[MCLE_OPT_TDL_KERNEL] == 4) return run_siso_tdl_wave_w<T, N, 4>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        else return run_siso_tdl_wave_w<T, N, 3>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
    } else {        // registers: 96 / 118 (256), 124 / 158 (512) in complex64 (stage twiddles included) / complex128
#ifndef MCLE_TDL_256_F32_WPS
#define MCLE_TDL_256_F32_WPS 6      // (six against five: +2.7 % with 9 spilled registers at the 80-register bound; seven: level)
#endif
#ifndef MCLE_TDL_256_F64_WPS
#define MCLE_TDL_256_F64_WPS 5      // (five against four: +2.6 % with 14 spilled at 96)
#endif
        constexpr int W = N == 256 ? (sizeof(T) == 8 ? MCLE_TDL_256_F64_WPS : MCLE_TDL_256_F32_WPS) : (sizeof(T) == 8 ? 3 : 4);
        return run_siso_tdl_wave_w<T, N, W>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
    }
}


}  // namespace mcle
