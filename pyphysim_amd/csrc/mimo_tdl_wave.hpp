// mimo_tdl_wave.hpp -- SURVEY.md section 8(f).1 (frequency-selective MIMO-OFDM) with ONE RECEIVE ANTENNA PER WAVEFRONT (round 5):
// k_run_mimo_ofdm_tdl_wave and its launcher; instantiated per arithmetic, size AND order mode in sixteen translation units
// (pipeline_mimo_tdl_wave_{f32,f64}_{256,512,1024,2048}[k].hip; k = the compile-time polynomial order of the benchmark -- at every size
// since the last day of round 6: the run-time-order kernels issue 1.4 x the vector and 5 x the scalar instructions per subcarrier).
//
// Reference path (restated by oracle/chains.py::chain_mimo_ofdm_tdl), as in pipeline_mimo_tdl.hip:
//   TdlMimoChannel / corrupt_data MIMO branch      channels/fading.py:1290-1333, :1092-1118
//   per-symbol mean frequency response             channels/fading.py:513-536 (+ modulators/ofdm.py:545-547)
//   Blast receive filter on every used bin         mimo/mimo.py:287-309, :577-607   (any Nr x Nt with Nt <= Nr)
//
// The kernel of rounds 1-4 (k_run_mimo_ofdm_tdl) shares every transform stage between the 256 threads of a workgroup: a dozen
// workgroup barriers per OFDM symbol, five LDS round trips per transform, delayed reads at swizzled addresses, and Nt = Nr only.
// Here a workgroup is Nr wavefronts and wavefront r IS receive antenna r (and transmit antenna r while r < Nt), the way
// config 3's wavefront kernel (siso_tdl_wave.hpp) made a wavefront a realization:
//   * transforms wave-local (fft_r16.hpp: radix-16 register passes at 1024, radix-4 stages at 256 / 512 / 2048): no barrier inside
//     a transform; FIVE workgroup barriers per symbol -- after the decode, after the scatter, after the transmit transforms (every
//     receive antenna reads every transmit antenna's time signal), inside the first receive pass between its arithmetic and its
//     stores (r16_pass BAR: the pass runs from registers while the other antennas finish reading), after the receive transforms;
//   * the transmit signals in natural order BEHIND THEIR CYCLIC PREFIX (hand-over through registers both ways): x_a[m - d] is
//     base(antenna, tap) + 64 c, immediate offsets, consecutive lanes on consecutive words; a tap beyond the prefix finds the
//     previous symbol's end in front of it (round 6: `s_hist`, written by the lane that held the sample);
//   * receive antenna r's S Nt tap polynomials parked across the lanes of wavefront r (the record of k_mimo_tdl_symbol_polys<T, true>:
//     register m / 2, lane 2 (s Nt + a) + m % 2) and read by v_readlane with a scalar lane index: the tap loop is a real loop
//     (code size: Nt x 16 samples x (2 K + 4) FMAs per trip), the delays sit in one register's lanes too;
//   * noise as in the config-3 kernel: one NOISE block per sample pair, lanes l and l + 1 draw half of the blocks each and swap
//     halves by DPP (ledger unchanged; an odd row start takes unpaired draws);
//   * the decode takes BQ subcarriers f0 + j N / BQ per work item, BQ in {1, 2} (static_assert below; a BQ = 4 form with
//     (-i)^(j d) classes was planned in the first draft and never built): w^((f0 + N / 2) d) = w^(f0 d) (-1)^d, so the S Nr Nt
//     products mean x twiddle are formed ONCE per work item, summed by the PARITY of their delay and spread over the two bins as
//     sum and difference -- H(f) costs 240 instructions per bin at 4 x 4 and five taps instead of 320 -- and the tap means are
//     read from LDS once per work item instead of once per bin;
//     H(f) is formed one receive antenna at a time and folded into the Gram matrix H^H H and H^H y (mimo.hpp: blast_gram_row /
//     blast_solve_gram), so that no thread ever holds BQ whole channel matrices.
// Same draw ledger (pipeline_mimo_tdl.hip) and the same arithmetic outside the transforms and H(f) (Horner per tap in tap order,
// Cholesky solve): complex128 per-realization counts equal the oracle's (tests/test_gpu_mimo_tdl_wave.py).
#pragma once
#include <type_traits>

#include "fft.hpp"
#include "fft_r16.hpp"
#include "mimo.hpp"
#include "mimo_tdl.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "pkcx.hpp"
#include "totals.hpp"
#include "walk_f64.hpp"
#include "wave_lanes.hpp"

namespace mcle {

constexpr int kMimoWaveMaxTaps = 8;

// position (after fft_dif) of bin j N / BQ -- the bits by which the BQ bins of a decode work item differ
template <int N, int BQ> constexpr int mimo_wave_posj(int j) {
    int f = j * (N / BQ), pos = 0, size = N;
    for (int s = 0; s < FftShape<N>::N4; ++s) {
        size >>= 2;
        pos += (f & 3) * size;
        f >>= 2;
    }
    if (FftShape<N>::HAS2) pos += (f & 1);
    return pos;
}
template <int N, int BQ> constexpr int mimo_wave_jmask() {
    int m = 0;
    for (int j = 0; j < BQ; ++j) m |= mimo_wave_posj<N, BQ>(j);
    return m;
}
// q -> the q-th position whose JM bits are clear
template <int N, int JM> __host__ __device__ __forceinline__ int mimo_wave_deposit(int q) {
    int p = 0, b = 0;
#pragma unroll
    for (int bit = 0; bit < FftShape<N>::LOG2; ++bit) {
        if ((JM >> bit) & 1) continue;
        p |= ((q >> b) & 1) << bit;
        ++b;
    }
    return p;
}
// base position of decode work item wi of lane ln.  1024 (swizzle lds_swz16f): the lane map of the planar kernel's fused stage
// (conflict free under the 32-lane read rule, tests/test_f64_layout.py); other sizes (lds_swz64): consecutive lanes on consecutive
// free positions (conflict free at 256, two-way at 512 / 2048 -- BQ reads per antenna and work item next to ~700 VALU instructions)
template <int N, int BQ> __host__ __device__ __forceinline__ int mimo_wave_p0(int ln, int wi) {
    if constexpr (N == 1024) {
        constexpr int EB = 4 / BQ;
        const int gq = wi / EB, e = wi % EB, h = (ln >> 5) & 1;
        const int g = (ln & 15) | (gq << 4) | (h << 6) | (((ln >> 4) & 1) << 7);
        return 4 * g + e;
    } else {
        return mimo_wave_deposit<N, mimo_wave_jmask<N, BQ>()>(ln + 64 * wi);
    }
}

// T, N: arithmetic, fft_size.  NT x NR: the geometry (NR wavefronts).  KT: polynomial order of the taps, compile time (> 0: the
// coefficients parked in (KT + 2) / 2 registers) or 0 = run time (coefficients fetched from the record by wave-uniform loads).
// BQ: subcarriers per decode work item (1, 2, 4).  WPS: wavefronts per SIMD the registers are bounded for.
// ABL (builds with -DMCLE_EXPERIMENTS only, option mimo_tdl_kernel = 16 + ABL): stage ablation for TIMING -- wrong results by
// construction: bit 0 the channel's tap loop, 1 the noise, 2 the Cholesky solves, 3 H(f) and the Gram rows, 4 the demodulator,
// 5 the four transform passes that are not fused with a hand-over, 6 the scatter's draws.
template <typename T, int N, int NT, int NR, int KT, int BQ, int WPS, int ABL = 0>
__global__ __launch_bounds__(64 * NR, WPS) void k_run_mimo_ofdm_tdl_wave(MimoTdlParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first,
                                                                         uint64_t count, const cx<T>* __restrict__ g_tw,
                                                                         const cx<T>* __restrict__ g_polys, mcle_counters* counters,
                                                                         uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    static_assert(NT >= 1 && NT <= NR && NR <= 4, "geometry");
    static_assert(N == 256 || N == 512 || N == 1024 || N == 2048, "fft_size");
    constexpr int R = N / 64;                                               // samples per lane
    constexpr bool R16 = N == 1024;
    constexpr int TB = 64 * NR;
    constexpr int NQK = KT > 0 ? (KT + 2) / 2 : 1;                          // parked registers (compile-time order)
    constexpr int WI = N / (64 * BQ);                                       // decode work items per symbol
    // the channel's delayed samples double-buffered: 2 x 32 more registers, which a complex64 wavefront has at two per SIMD
    constexpr bool CHAN_DB = sizeof(T) == 4 && KT > 0 && WPS <= 2;
    static_assert(WI >= 1, "BQ");
    auto swz = [](int e) { return R16 ? lds_swz16f(e) : lds_swz64(e); };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tid = threadIdx.x;
    const int S = pp.n_taps, K = KT > 0 ? KT : pp.K;
    const int U = pp.num_used, cp = pp.cp, W = N + cp;
    const int PS = S * NR * NT;
    const int NQ = mimo_tdl_nq(K), LW = 2 * S * NT;
    const int n_coef = NR * NQ * LW, rec_len = n_coef + PS;
    // A wavefront's sample memory: two planes of `pitch` = N + P scalars (P = the largest tap delay rounded up to 16).  During the
    // transforms a plane holds the N swizzled elements; between them it is the time signal WITH ITS CYCLIC PREFIX in natural
    // order -- xp[P + m] = x[m], xp[j] = x[N - P + j] -- read by every receive antenna at lane-linear addresses.
    const int pitch = pp.x_elems, P = pitch - N;
    T* s_all = reinterpret_cast<T*>(smem);                                   // [NR][2][pitch]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_all + NR * 2 * pitch);      // [M rounded to 2]
    cx<T>* s_txtab = s_table + ((mp.M + 1) & ~1);                           // ... x the transmit scale
    cx<T>* s_mean = s_txtab + ((mp.M + 1) & ~1);                            // [PS rounded to 2] the symbol's tap means
    // the candidate grid lives in LDS only where it is hot: without a certificate (8- / 16-PSK ...) every decision goes through it;
    // with one the table search serves ~4 eps of the symbols and reads the grid from the L2-resident copy
    const bool grid_lds = mp.cert == 0 && mp.grid.G > 0;
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_mean + ((PS + 1) & ~1));
    const int grid_words = grid_lds ? ((mp.grid.G * mp.grid.G + 1) & ~1) : 0;
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + grid_words);        // [NT U rounded to 16]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_idx + ((NT * U + 15) & ~15));         // [2][4][2]
    // A tap beyond the cyclic prefix (inter-symbol interference, round 6): plane positions [0, P - cp) then hold the END OF THE
    // PREVIOUS SYMBOL -- sample m - d < -cp of the stream is x_prev[N + m - d + cp] -- kept per transmit antenna in `hist`
    // between the symbols (zeros in front of the first one, channels/fading.py:1092-1118: the filter starts empty)
    const bool isi = pp.dmax > cp;
    const int HL = isi ? pitch - N - cp : 0;
    cx<T>* s_hist = reinterpret_cast<cx<T>*>(s_part + 16);                  // [NT][HL]
    T* pr = s_all + w * 2 * pitch;                                         // transform planes of this antenna: re [0, N), im [N, 2 N)
    T* pi = pr + N;
    cx<T>* xp = reinterpret_cast<cx<T>*>(pr);                              // natural-order signal with prefix, (re, im) interleaved:
                                                                            // [0, pitch) -- the same memory, never live together
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];             // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, tid, TB);
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NT) / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const T nv_filter = (T)(pp.mmse ? pp.noise_var : 0.0);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    for (int m = tid; m < mp.M; m += TB) {
        const cx<T> c = mp.g_table[m];
        s_table[m] = c;
        s_txtab[m] = cscale(c, tx_scale);
    }
    if (grid_lds) load_grid(mp, s_grid);
    const unsigned long long* gridp = grid_lds ? s_grid : mp.grid.cells;
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    const int per_sym = U * NT;
    const uint64_t n_total = (uint64_t)pp.n_ofdm_sym * W;                   // samples per antenna
    const uint64_t noise_row = n_total + (uint64_t)pp.dmax;                 // noise is drawn for the whole faded stream
    const double xc = 0.5 * (double)(W - 1);                                // centre of the symbol in local sample units
    // tap delays: in registers for the unrolled loops of the decode, across the lanes of one register for the channel's real loop
    int dly[kMimoWaveMaxTaps];
    int dlyv = 0;
#pragma unroll
    for (int s = 0; s < kMimoWaveMaxTaps; ++s) {
        dly[s] = s < S ? pp.tap_delay[s] : 0;
        if (lane == s) dlyv = dly[s];
    }
    [[maybe_unused]] int clsv = -1;                                          // lane p < 8: class position p of the decode (mimo_tdl.hpp)
    if constexpr (BQ == 2) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (lane == p) clsv = pp.cls_code[p];
    }
    [[maybe_unused]] R16Tw64<T> tw16;
    // H(f) at two bins per lane by delay-class positions: complex64 +3 % at 1024 4x4, complex128 +2.5 % (once its multiply-add was the
    // chained cfma4: level before, and two set-up registers spilled at 4x4 -- stored once, reloaded once per symbol);
    // profiles/r05/f1_hf_class_ab.log
    constexpr bool CLS2 = BQ == 2 && !(ABL & 256);
    constexpr int REP4 = N / 256;
    // radix-4 sizes, complex64: the lane's stage twiddles in registers at 256 (24 registers; the 48 of 512 spilled 36 registers
    // next to the Gram accumulators of the decode)
    constexpr bool WTW = !R16 && sizeof(T) == 4 && REP4 * FftShape<N>::N4 * 6 <= 24;
    [[maybe_unused]] TwRegs64<T, N> twr4[WTW ? REP4 : 1];
    if constexpr (WTW) {
#pragma unroll
        for (int rep = 0; rep < REP4; ++rep) twr4[rep] = load_tw64<T, N>(g_tw, lane + 64 * rep);
    }
    const bool slicer = mp.method == MCLE_DEMOD_QAM_SLICER;
    const bool certpath = !slicer && mp.cert != 0;
    const int hU = U / 2;

    uint64_t it = 0, rl_prev = 0;
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x, ++it) {
        const Rng rng(seed, first + rl);
        const int buf = (int)(it & 1);
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            const uint64_t sym0 = (uint64_t)os * W;
            int gi = opaque(lane);
            // the symbol's record: this antenna's tap polynomials parked across the lanes (compile-time order), fetched here and
            // first used after the transmit transform
            const cx<T>* __restrict__ g_rec = g_polys + (rl * pp.n_ofdm_sym + os) * (uint64_t)rec_len;
            [[maybe_unused]] cx<T> prk[NQK];
            if constexpr (KT > 0) {
#pragma unroll
                for (int q = 0; q < NQK; ++q) prk[q] = gi < LW ? g_rec[(w * NQK + q) * LW + gi] : mk<T>(0, 0);
            }
            __syncthreads();                                                // B0: the previous symbol's decode has read the planes
            if (tid == 0 && os == 0 && it > 0) {                            // every wave is past the previous realization: account it
                const unsigned* q = s_part + (buf ^ 1) * 8;
                unsigned ts = 0, tb = 0;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    ts += q[2 * i];
                    tb += q[2 * i + 1];
                }
                wg_account(totals, ts, tb, false, rl_prev, sym_out, bit_out);
            }
            const int t0 = opaque(tid);     // (addresses derived from it are recomputed here, not parked across the realization loop)
            for (int e = t0; e < PS; e += TB) s_mean[e] = g_rec[n_coef + e];             // first read in the decode
            // ---- transmit: symbols -> bins at digit-reversed positions of the transmit antennas' planes ----
            if (U != N) {
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    for (int p = t0; p < 2 * N; p += TB) s_all[a * 2 * pitch + p] = 0;
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * per_sym, n_last = n_first + per_sym;
            const bool aligned_scatter = (16 % NT == 0) && U == N && (per_sym & 15) == 0;
            for (uint64_t blk = (n_first >> 4) + t0; blk <= ((n_last - 1) >> 4); blk += TB) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (aligned_scatter) {          // a block = the NT antennas of 16 / NT consecutive subcarriers d0 + t: bin(d0 + t) =
                    const int nl0 = (int)((blk << 4) - n_first);             // bin(d0) ^ t, digit reversal and swizzle XOR-linear
                    const int p0 = swz(fft_pos_of_index<N>(ofdm_bin(nl0 / NT, N, U)));
                    *reinterpret_cast<uint4*>(s_idx + nl0) = make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u),
                                                                        dw.w[2] & (mask * 0x01010101u), dw.w[3] & (mask * 0x01010101u));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const cx<T> c = s_txtab[tx];
                        const int pos = p0 ^ swz(fft_pos_of_index<N>(j / NT));
                        T* pa = s_all + (j % NT) * 2 * pitch;
                        pa[pos] = c.x;
                        pa[N + pos] = c.y;
                    }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t n = (blk << 4) + j;
                    if (n >= n_first && n < n_last) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const int nl = (int)(n - n_first);
                        const int a = nl % NT, d = nl / NT;
                        s_idx[nl] = (unsigned char)tx;
                        const cx<T> c = s_txtab[tx];
                        const int pos = swz(fft_pos_of_index<N>(ofdm_bin(d, N, U)));
                        T* pa = s_all + a * 2 * pitch;
                        pa[pos] = c.x;
                        pa[N + pos] = c.y;
                    }
                }
            }
            if constexpr (R16) tw16 = load_r16_tw<T>(g_tw, opaque(lane));    // in flight across the barrier
            __syncthreads();                                                // B1: every antenna's bins are in place
            // ---- transmit transform (wavefront a = transmit antenna a): time samples -> natural order behind the prefix ----
            cx<T> y[R];                                                     // element gi + 64 c in y[c]
            if (w < NT) {
                if constexpr (R16) {
                    r16_dit<T, true, true, false, true>(pr, pi, lane, tw16, g_tw, y);   // the last pass leaves the samples in registers
                } else {
                    wave_fft_dit<T, N, true, WTW>(pr, g_tw, lane, twr4);
                    gi = opaque(lane);
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int sl = swz(gi) ^ swz(64 * c);
                        y[c] = mk<T>(pr[sl], pi[sl]);
                    }
                }
                r16_wave_sync();                                            // every lane's reads of the planes are issued
                gi = opaque(lane);
#pragma unroll
                for (int c = 0; c < R; ++c) xp[P + gi + 64 * c] = y[c];
#pragma unroll
                for (int c = (R > 4 ? R - 4 : 0); c < R; ++c)               // the prefix: the last P samples once more (P <= 256)
                    if (gi + 64 * c >= N - P + HL) xp[gi + 64 * c - (N - P)] = y[c];
                if (isi) {                                                  // in front of the prefix: the previous symbol's end;
                    cx<T>* hist = s_hist + w * HL;                          // the lane that holds sample e now held it then
#pragma unroll
                    for (int c = (R > 4 ? R - 4 : 0); c < R; ++c) {
                        const int j = gi + 64 * c - (N - HL);
                        if (j >= 0) {
                            xp[j] = os > 0 ? hist[j] : mk<T>(0, 0);
                            hist[j] = y[c];
                        }
                    }
                }
            }
            __syncthreads();                                                // B2: every transmit signal is in place
            // ---- channel: y_r[m] = sum_s sum_a g_sra(x) x_a[m - d_s] for this lane's R samples m = gi + 64 c.  The polynomials of this
            //      record are expanded around the OUTPUT sample (k_mimo_tdl_symbol_polys<T, true>: tap s about the symbol centre minus
            //      d_s samples), so one abscissa x = cp + m - xc serves every tap: exact (half-)integers below 2^12, in float too ----
            gi = opaque(lane);
#pragma unroll
            for (int c = 0; c < R; ++c) y[c] = mk<T>(0, 0);
            {
                const T x0 = (T)(2 * (cp + gi) - (W - 1)) * (T)0.5;        // (cp + gi) - xc from integers: exact, nothing carried
                const int NP = (ABL & 1) ? 0 : S * NT;                      // (tap, transmit antenna) pairs, p = s NT + a
                const cx<T>* xbase = reinterpret_cast<const cx<T>*>(s_all) + (P + gi);
                // Sixteen samples of the lane at a time (the whole lane at <= 1024 points; two rounds of the pair loop at 2048, where
                // 32 accumulators + 32 delayed samples + their abscissae did not fit the registers and the accumulators were
                // spilled inside the loop)
                constexpr int CHN = R < 16 ? R : 16;
                static_for<R / CHN>([&](auto hc) {
                    constexpr int C0 = decltype(hc)::value * CHN;
                    // the CHN delayed samples of pair p: x_a[m - d] = xd[64 c], d <= P -- one address, immediate offsets
                    auto load_pair = [&](int p, cx<T> (&buf)[CHN]) {
                        const int s = p / NT, a = p - s * NT;
                        const cx<T>* xd = xbase + (a * pitch - __builtin_amdgcn_readlane(dlyv, s));
#pragma unroll
                        for (int c = 0; c < CHN; ++c) {
                            if constexpr (ABL & 64) buf[c] = mk<T>((T)(size_t)xd, x0);   // (timing: the channel stage without its LDS reads)
                            else buf[c] = xd[64 * (C0 + c)];
                        }
                    };
                    auto mac_pair = [&](int p, const cx<T> (&buf)[CHN]) {
                        if constexpr (KT > 0) {
                            cx<T> cc[KT + 1];
#pragma unroll
                            for (int m = 0; m <= KT; ++m)
                                cc[m] = mk<T>(lane_value(prk[m >> 1].x, 2 * p + (m & 1)), lane_value(prk[m >> 1].y, 2 * p + (m & 1)));
#pragma unroll
                            for (int c = 0; c < CHN; ++c)
                                chan_step<KT, !(ABL & 128)>(y[C0 + c], cc, x0 + (T)(64 * (C0 + c)) /* exact */, buf[c]);
                        } else {                // run-time order: the coefficients by wave-uniform loads from the record, four
                            const cx<T>* __restrict__ cb = g_rec + (size_t)w * NQ * LW + 2 * p;   // samples at a time (registers)
                            constexpr int CH = CHN < 4 ? CHN : 4;
#pragma unroll
                            for (int c0 = 0; c0 < CHN; c0 += CH) {
                                cx<T> g[CH];
                                const cx<T> top = cb[(K >> 1) * LW + (K & 1)];
#pragma unroll
                                for (int c = 0; c < CH; ++c) g[c] = top;
                                for (int mm = K - 1; mm >= 0; --mm) {
                                    const cx<T> cm = cb[(mm >> 1) * LW + (mm & 1)];
#pragma unroll
                                    for (int c = 0; c < CH; ++c) {
                                        const T xx = x0 + (T)(64 * (C0 + c0 + c));
                                        g[c].x = fma(g[c].x, xx, cm.x);
                                        g[c].y = fma(g[c].y, xx, cm.y);
                                    }
                                }
#pragma unroll
                                for (int c = 0; c < CH; ++c) y[C0 + c0 + c] = cfma4(g[c], buf[c0 + c], y[C0 + c0 + c]);
                            }
                        }
                    };
                    if constexpr (CHAN_DB) {    // two sample buffers: the loads of pair p + 1 fly while pair p is consumed
                        cx<T> xa[CHN], xb[CHN];
                        if (NP > 0) load_pair(0, xa);
                        int p = 0;
#pragma nounroll
                        for (; p + 1 < NP; p += 2) {
                            load_pair(p + 1, xb);
                            mac_pair(p, xa);
                            if (p + 2 < NP) load_pair(p + 2, xa);
                            mac_pair(p + 1, xb);
                        }
                        if (p < NP) mac_pair(p, xa);
                    } else {
#pragma nounroll
                        for (int p = 0; p < NP; ++p) {
                            cx<T> xa[CHN];
                            load_pair(p, xa);
                            mac_pair(p, xa);
                        }
                    }
                });
            }
            if constexpr (R16) tw16 = load_r16_tw<T>(g_tw, opaque(lane));    // in flight behind the noise draws
            // ---- noise: sample w noise_row + sym0 + cp + m of the NOISE stream ----
            const uint64_t nbase = (uint64_t)w * noise_row + sym0 + (uint64_t)cp;
            if constexpr (ABL & 2) {
            } else if ((nbase & 1) == 0) {      // lanes l (even), l + 1 share the block of samples m, m + 1
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
                    const uint32_t x0w = (i0 & 1) ? b.w[2] : b.w[0], x1w = (i0 & 1) ? b.w[3] : b.w[1];
                    cx<T> z;
                    if constexpr (sizeof(T) == 8) z = cn_from_words_lds(x0w, x1w, sigma, s_bm);
                    else z = cn_from_words(x0w, x1w, sigma);
                    y[c] = cadd(y[c], z);
                }
            }
            // ---- receive transform; B3 (every antenna has read every transmit signal) sits between the first pass's arithmetic
            //      and its stores ----
            if constexpr (R16) {
                r16_dif<T, false, true, false, true, true>(pr, pi, lane, tw16, g_tw, y);   // bins at digit-reversed positions
            } else {
                __syncthreads();                                            // B3
                gi = opaque(lane);
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int sl = swz(gi) ^ swz(64 * c);
                    pr[sl] = y[c].x;
                    pi[sl] = y[c].y;
                }
                r16_wave_sync();
                wave_fft_dif<T, N, false, WTW>(pr, g_tw, lane, twr4);
            }
            __syncthreads();                                                // B4: every receive antenna's bins are in place
            // ---- receive: per work item BQ subcarriers f0 + j N / BQ -- H(f) from the tap means, the Gram matrix row by row,
            //      Cholesky solve, demodulate, count ----
            const int ln = opaque(lane);
            for (int wi = w; wi < WI; wi += NR) {
                const int p0 = mimo_wave_p0<N, BQ>(ln, wi);
                const int slot0 = swz(p0);
                const int f0 = fft_index_of_pos<N>(p0);
                cx<T> Wt[kMimoWaveMaxTaps];                                  // w^(f0 d_s)
                [[maybe_unused]] int ptap[kMimoWaveMaxTaps];                 // BQ == 2: the tap at class position s
#pragma unroll
                for (int s = 0; s < kMimoWaveMaxTaps; ++s) {
                    if constexpr (ABL & 2048) {
                        Wt[s] = mk<T>((T)(f0 + s) * (T)1e-3, (T)(f0 - s) * (T)1e-3);      // (timing: no gathers)
                    } else if constexpr (CLS2) {                         // class positions: w^(f0 d) of the tap there
                        const int code = __builtin_amdgcn_readlane(clsv, s);
                        ptap[s] = code >> 16;
                        Wt[s] = code >= 0 ? g_tw[(f0 * (code & 0xffff)) & (N - 1)] : mk<T>(0, 0);
                    } else {
                        Wt[s] = s < S ? g_tw[(f0 * dly[s]) & (N - 1)] : mk<T>(0, 0);
                    }
                }
                uint32_t sent[BQ];
                bool valid[BQ];
#pragma unroll
                for (int j = 0; j < BQ; ++j) {
                    const int f = f0 + j * (N / BQ);
                    int d;                                                  // data position of bin f (inverse of ofdm_bin)
                    if (U == N) {
                        d = (f + N / 2) & (N - 1);
                        valid[j] = true;
                    } else {
                        const bool neg = f >= N - hU, pos = f >= 1 && f <= hU;
                        d = neg ? f - (N - hU) : (pos ? hU + f - 1 : 0);
                        valid[j] = neg || pos;
                    }
                    if constexpr (NT == 4) {
                        sent[j] = *reinterpret_cast<const uint32_t*>(s_idx + 4 * d);
                    } else if constexpr (NT == 2) {
                        sent[j] = *reinterpret_cast<const uint16_t*>(s_idx + 2 * d);
                    } else {
                        sent[j] = 0;
#pragma unroll
                        for (int a = 0; a < NT; ++a) sent[j] |= (uint32_t)s_idx[NT * d + a] << (8 * a);
                    }
                }
                cx<T> A[BQ][NT][NT], b[BQ][NT];
#pragma unroll
                for (int j = 0; j < BQ; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        b[j][i] = mk<T>(0, 0);
#pragma unroll
                        for (int k = 0; k < NT; ++k) A[j][i][k] = mk<T>(0, 0);
                    }
#pragma unroll
                for (int r = 0; r < ((ABL & 8) ? 0 : NR); ++r) {
                    static_assert(BQ == 1 || BQ == 2, "a lane decodes one bin or the pair f0, f0 + N / 2");
                    cx<T> u[BQ][NT];                                        // row r of H at the lane's BQ bins
#pragma unroll
                    for (int c = 0; c < BQ; ++c)
#pragma unroll
                        for (int a = 0; a < NT; ++a) u[c][a] = mk<T>(0, 0);
                    if constexpr (CLS2) {
                        // two bins per lane: ONE complex multiply-add per entry and tap into the tap's delay class, the bins are
                        // the sum and the difference of the classes (u[0] = even delays, u[1] = odd delays until the butterfly)
                        const auto into = [&](cx<T>* uc, int pos) {
                            const int tap = ptap[pos];
#pragma unroll
                            for (int a = 0; a < NT; ++a) {
                                const cx<T> m = s_mean[(tap * NR + r) * NT + a];
                                if constexpr (sizeof(T) == 4) uc[a] = from_pk(pk_cfma(to_pk(m), to_pk(Wt[pos]), to_pk(uc[a])));
                                else uc[a] = cfma4(m, Wt[pos], uc[a]);
                            }
                        };
#pragma unroll
                        for (int k = 0; k < kMimoWaveMaxTaps; ++k) {
                            if (k >= pp.cls_ne) break;
                            into(u[0], k);
                        }
#pragma unroll
                        for (int k = 0; k < kMimoWaveMaxTaps; ++k) {
                            if (k >= pp.cls_no) break;
                            into(u[1], kMimoWaveMaxTaps - 1 - k);
                        }
#pragma unroll
                        for (int a = 0; a < NT; ++a) {
                            const cx<T> e = u[0][a];
                            u[0][a] = cadd(e, u[1][a]);
                            u[1][a] = csub(e, u[1][a]);
                        }
                    } else
#pragma unroll
                    for (int s = 0; s < kMimoWaveMaxTaps; ++s) {
                        if (s >= S) break;
                        if constexpr (BQ == 1) {                            // one bin per lane: a complex multiply-add per entry and tap
#pragma unroll
                            for (int a = 0; a < NT; ++a) {
                                const cx<T> m = s_mean[(s * NR + r) * NT + a];
                                if constexpr (sizeof(T) == 4 && !(ABL & 128)) u[0][a] = from_pk(pk_cfma(to_pk(m), to_pk(Wt[s]), to_pk(u[0][a])));
                                else u[0][a] = cfma4(m, Wt[s], u[0][a]);
                            }
                        } else {
                            // two bins per lane, the sign form (experiments: ABL & 256): t = mean x twiddle once, H(f0) += t,
                            // H(f0 + N / 2) += (-1)^d t -- no select, no branch
                            cx<T> t[NT];
#pragma unroll
                            for (int a = 0; a < NT; ++a) {
                                if constexpr (ABL & 512) t[a] = from_pk(pk_cmul(to_pk(Wt[(s + a + r) % 5]), to_pk(Wt[s])));   // (timing: no LDS reads of the means)
                                else if constexpr (sizeof(T) == 4 && !(ABL & 128)) t[a] = from_pk(pk_cmul(to_pk(s_mean[(s * NR + r) * NT + a]), to_pk(Wt[s])));
                                else t[a] = cmul(s_mean[(s * NR + r) * NT + a], Wt[s]);
                            }
                            int dl = dly[s];
                            asm volatile("" : "+s"(dl));                    // (recomputed here: hoisted, the eight sign pairs were spilled)
                            const T sg = (dl & 1) ? (T)-1 : (T)1;
#pragma unroll
                            for (int a = 0; a < NT; ++a) {
                                u[0][a] = cadd(u[0][a], t[a]);
                                u[1][a].x = fma(sg, t[a].x, u[1][a].x);
                                u[1][a].y = fma(sg, t[a].y, u[1][a].y);
                            }
                        }
                    }
                    const T* prr = s_all + r * 2 * pitch;
#pragma unroll
                    for (int j = 0; j < BQ; ++j) {
                        const int sl = slot0 ^ swz(mimo_wave_posj<N, BQ>(j));
                        if constexpr (ABL & 1024) blast_gram_row<T, NT>(u[j], mk<T>((T)sl, (T)(sl + r)), A[j], b[j]);   // (timing: no LDS reads of the bins)
                        else blast_gram_row<T, NT>(u[j], mk<T>(prr[sl], prr[N + sl]), A[j], b[j]);
                    }
                }
                cx<T> est[BQ * NT];
#pragma unroll
                for (int j = 0; j < BQ; ++j) {
                    cx<T> x[NT];
                    bool ok = true;
                    if constexpr (ABL & 4) {
#pragma unroll
                        for (int a = 0; a < NT; ++a) x[a] = cadd(b[j][a], A[j][a][0]);
                    } else {
                        ok = blast_solve_gram<T, NT>(A[j], nv_filter, b[j], x);            // filter applied, never formed
                    }
#pragma unroll
                    for (int a = 0; a < NT; ++a) est[j * NT + a] = ok ? cscale(x[a], rx_scale) : mk<T>(0, 0);   // singular: ZF only
                }
                // Four streams and a square QAM: the bin's four decisions counted in the LEVEL domain (round 6) -- complex64 by the packed
                // slicer of the planar kernels (qam_pack.hpp: four decisions per v_cvt_pk_u8_f32 word, ~3 instructions per symbol instead
                // of ~25), complex128 by walk_qam_count4 (walk_f64.hpp: slicer or margin certificate, ~18 instead of ~27 - 36)
                bool counted = false;
                if constexpr (NT == 4 && !(ABL & 16)) {
                    if constexpr (sizeof(T) == 4) {
                        if (slicer) {
                            const QamPack qp = qam_pack(mp);
#pragma unroll
                            for (int j = 0; j < BQ; ++j) {
                                const f4q er = {est[4 * j].x, est[4 * j + 1].x, est[4 * j + 2].x, est[4 * j + 3].x};
                                const f4q ei = {est[4 * j].y, est[4 * j + 1].y, est[4 * j + 2].y, est[4 * j + 3].y};
                                const uint32_t x = valid[j] ? (qam_levels4(er, ei, qp) ^ labels_to_levels(sent[j], qp)) : 0u;
                                qam_count4(x, qp, se, be);
                            }
                            counted = true;
                        } else if (certpath && mp.cert == 1) {      // the margin certificate in the level domain too (last day of round 6)
#pragma unroll
                            for (int j = 0; j < BQ; ++j) {
                                const cx<T> e4[4] = {est[4 * j], est[4 * j + 1], est[4 * j + 2], est[4 * j + 3]};
                                if (valid[j]) walk_qam_count4<T, true>(mp, s_table, e4, sent[j], se, be);
                            }
                            counted = true;
                        }
                    } else {
                        if (slicer || (certpath && mp.cert == 1)) {
#pragma unroll
                            for (int j = 0; j < BQ; ++j) {
                                const cx<T> e4[4] = {est[4 * j], est[4 * j + 1], est[4 * j + 2], est[4 * j + 3]};
                                if (valid[j]) {                  // (a bin outside the band holds no symbols: nothing to certify either)
                                    if (slicer) walk_qam_count4<false>(mp, s_table, e4, sent[j], se, be);
                                    else walk_qam_count4<true>(mp, s_table, e4, sent[j], se, be);
                                }
                            }
                            counted = true;
                        }
                    }
                }
                if (!counted) {
                int dec[BQ * NT];
                if constexpr (ABL & 16) {
#pragma unroll
                    for (int i = 0; i < BQ * NT; ++i) dec[i] = (int)est[i].x & 63;
                } else if (slicer) {
#pragma unroll
                    for (int i = 0; i < BQ * NT; ++i) dec[i] = demod_qam_slicer<T>(est[i], mp.qam_scale, mp.qam_L, mp.half_bits);
                } else if (certpath) {
                    bool unsure = false;
#pragma unroll
                    for (int i = 0; i < BQ * NT; ++i) {
                        bool sure;
                        dec[i] = demod_cert_any<T>(mp, est[i], sure);
                        unsure = unsure || (valid[i / NT] && !sure);
                    }
                    if (unsure) {
#pragma unroll
                        for (int i = 0; i < BQ * NT; ++i) dec[i] = demod_one(mp, s_table, gridp, est[i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < BQ * NT; ++i) dec[i] = demod_one(mp, s_table, gridp, est[i]);
                }
#pragma unroll
                for (int i = 0; i < BQ * NT; ++i) {
                    const unsigned x = valid[i / NT] ? (((sent[i / NT] >> (8 * (i % NT))) & 0xFFu) ^ (unsigned)dec[i]) : 0u;
                    se += (x != 0u);
                    be += __popc(x);
                }
                }
            }
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        if (lane == 0) {
            s_part[buf * 8 + 2 * w] = se;
            s_part[buf * 8 + 2 * w + 1] = be;
        }
        rl_prev = rl;
    }
    __syncthreads();
    if (tid == 0) {
        if (it > 0) {
            const unsigned* q = s_part + (int)((it - 1) & 1) * 8;
            unsigned ts = 0, tb = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                ts += q[2 * i];
                tb += q[2 * i + 1];
            }
            wg_account(totals, ts, tb, false, rl_prev, sym_out, bit_out);
        }
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym, (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
    }
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (the caller goes on to the kernel of rounds 1-4 or
// reports the configuration as one for the staged operator chain)
template <typename T, int N, int NT, int NR, int KT, int BQ, int WPS, int ABL = 0>
int launch_mimo_tdl_wave(mcle_ctx* ctx, const MimoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                         mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, method);
    MimoTdlParams pw = pp;
    pw.x_elems = N + ((pp.dmax + 15) & ~15);                                // plane pitch: N + the prefix the taps reach into
    const int S = pp.n_taps, PS = S * NR * NT;
    const bool grid_lds = mp.cert == 0 && mp.grid.G > 0;
    const size_t lds = (size_t)NR * 2 * pw.x_elems * sizeof(T) + 2 * (((size_t)mp.M + 1) & ~(size_t)1) * sizeof(cx<T>) +
                       (((size_t)PS + 1) & ~(size_t)1) * sizeof(cx<T>) +
                       (grid_lds ? (((size_t)mp.grid.G * mp.grid.G + 1) & ~(size_t)1) * sizeof(unsigned long long) : 0) +
                       (((size_t)NT * pp.num_used + 15) & ~(size_t)15) + 16 * sizeof(unsigned) +
                       (pp.dmax > pp.cp ? (size_t)NT * (pw.x_elems - N - pp.cp) * sizeof(cx<T>) : 0);   // the previous symbol's end
    const size_t lds_static = (sizeof(T) == 8 ? (size_t)kBmLdsDoubles * 8 : 8) + sizeof(WgTotals) + 64;
    if (lds + lds_static > (size_t)160 * 1024) return MCLE_E_UNSUPPORTED;
    auto kern = k_run_mimo_ofdm_tdl_wave<T, N, NT, NR, KT, BQ, WPS, ABL>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + lds_static + 256));
    const int by_waves = WPS * 4 / NR;                                      // what __launch_bounds__ allocated registers for
    if (per_cu < 1) per_cu = 1;
    if (per_cu > by_waves) per_cu = by_waves;
    const size_t rec_len = mimo_tdl_wave_rec(S, NT, NR, pp.K);
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * rec_len;             // complex values per realization
    // records of a slice of realizations (k_mimo_tdl_symbol_polys<T, true>), then their links: <= 4 GiB of records per pair of
    // launches (3.2 / 9 kB per realization and symbol at 4 x 4 and five taps), so that a bench step of 393 216 realizations is ONE
    // dispatch of each kernel -- the 256 MiB slices of rounds 3-4 cut a step into unequal dispatches, which is what the per-launch
    // means of the round-4 profiles got wrong; shorter launches were never faster (pipeline_mimo_tdl.hip)
    // (a crowded or smaller device gets a smaller slice instead of an out-of-memory error: scratch_upto halves the request down to
    // 64 realizations' worth)
    uint64_t slice = (4096ull << 20) / (per_real * sizeof(cx<T>));
    if (slice < 1) slice = 1;
    if (slice > count) slice = count;
    void* recs = nullptr;
    size_t got = 0;
    const size_t one = (size_t)per_real * sizeof(cx<T>);
    if ((rc = ctx->scratch_upto((size_t)slice * one, (slice < 64 ? slice : 64) * one, &recs, &got))) return rc;
    if (got / one < slice) slice = got / one;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        launch_mimo_tdl_symbol_polys<T, true>(ctx->stream, pw, PS, NR * NT, N + pp.cp, seed, first + off, n, (cx<T>*)recs, NT);
        MCLE_LAUNCH_CHECK();
        // (realizations per workgroup: >= 8, and enough for the counters' flush to disappear at the small shapes -- pipe_common.hpp;
        //  (256, 2 x 2) complex64: 1.28 -> 1.5e8)
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, (uint64_t)ctx->n_cu * per_cu, n,
                                                            flush_min_units(8, sizeof(T) == 4 ? 12800 : 6400, (uint64_t)N * NR), 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NR), lds, ctx->stream, pw, mp, seed, first + off, n, (const cx<T>*)tw,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// wavefronts per SIMD the registers are bounded for (= workgroups of 4 wavefronts per CU the LDS admits at Nr = 4): 1024 and below
// complex64 3 / complex128 2; 2048: 2 / 1.  Subcarriers per decode work item: 2 where a wavefront has at least two.
// (complex64 at 1024: three -- 4.47 ms per 83 886 realizations with 12 spilled registers against 4.89 at two, where nothing spills
//  and the channel's delayed samples are double-buffered, and 4.91 at four: scripts/experiments/r05_calls.txt [call 7])
// (round 6: at 256 points the LDS admits more workgroups than that bound and the kernels fit 167 / 127 registers: three / four
//  wavefronts per SIMD there; at 512 the tighter bounds spill 92 / 29 registers and stay (profiles/r06/f1_family_rates.log))
#ifndef MCLE_MIMO_TDL_WPS_256_F64
#define MCLE_MIMO_TDL_WPS_256_F64 3
#endif
#ifndef MCLE_MIMO_TDL_WPS_256_F32
#define MCLE_MIMO_TDL_WPS_256_F32 4
#endif
// 512 points: ONE bin per decode work item frees the registers for one more wavefront per SIMD -- complex128 three instead of two:
// 1.09 -> 1.39e7 realizations/s (two bins at three wavefronts spill 92 registers); complex64: MCLE_MIMO_TDL_512_F32_BQ1 (A/B)
#ifndef MCLE_MIMO_TDL_512_F64_BQ1
#define MCLE_MIMO_TDL_512_F64_BQ1 1
#endif
#ifndef MCLE_MIMO_TDL_512_F32_BQ1
#define MCLE_MIMO_TDL_512_F32_BQ1 1
#endif
#ifndef MCLE_MIMO_TDL_1024_F32_BQ1
#define MCLE_MIMO_TDL_1024_F32_BQ1 0     // complex64 at 1024: one bin per work item at four wavefronts per SIMD (A/B: 2.026 against 2.033e7, not adopted)
#endif
template <typename T, int N> constexpr int mimo_tdl_wave_wps() {
    if (N == 1024 && sizeof(T) == 4 && MCLE_MIMO_TDL_1024_F32_BQ1) return 4;
    if (N == 512 && sizeof(T) == 8 && MCLE_MIMO_TDL_512_F64_BQ1) return 3;
    if (N == 512 && sizeof(T) == 4 && MCLE_MIMO_TDL_512_F32_BQ1) return 4;
    if (N <= 256) return sizeof(T) == 8 ? MCLE_MIMO_TDL_WPS_256_F64 : MCLE_MIMO_TDL_WPS_256_F32;
    return N >= 2048 ? (sizeof(T) == 8 ? 1 : 2) : (sizeof(T) == 8 ? 2 : 3);
}
template <int N, int NR> constexpr int mimo_tdl_wave_bq() { return N / (64 * NR) >= 2 ? 2 : 1; }
template <typename T, int N, int NR> constexpr int mimo_tdl_wave_bq_t() {
    if (N == 1024 && sizeof(T) == 4 && MCLE_MIMO_TDL_1024_F32_BQ1) return 1;
    return (N == 512 && ((sizeof(T) == 8 && MCLE_MIMO_TDL_512_F64_BQ1) || (sizeof(T) == 4 && MCLE_MIMO_TDL_512_F32_BQ1))) ? 1 : mimo_tdl_wave_bq<N, NR>();
}
// the polynomial order whose coefficients are parked in registers (the order of the benchmark's Doppler in each arithmetic);
// every other order runs the run-time-order kernels (KT = 0)
template <typename T> constexpr int mimo_tdl_wave_kf() { return sizeof(T) == 8 ? 5 : 2; }

// one size and one order mode: every 1 <= Nt <= Nr <= 4
template <typename T, int N, int KT>
int run_mimo_tdl_wave_size(mcle_ctx* ctx, int nt, int nr, const MimoTdlParams& pp, int method, uint64_t seed, uint64_t first,
                           uint64_t count, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (KT > 0 && pp.K != KT) return MCLE_E_UNSUPPORTED;
#define MCLE_WAVE_GEOM(NT_, NR_)                                                                                              \
    if (nt == NT_ && nr == NR_)                                                                                               \
        return launch_mimo_tdl_wave<T, N, NT_, NR_, KT, mimo_tdl_wave_bq_t<T, N, NR_>(), mimo_tdl_wave_wps<T, N>()>(          \
            ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
    MCLE_WAVE_GEOM(1, 1) MCLE_WAVE_GEOM(1, 2) MCLE_WAVE_GEOM(2, 2) MCLE_WAVE_GEOM(1, 3) MCLE_WAVE_GEOM(2, 3) MCLE_WAVE_GEOM(3, 3)
    MCLE_WAVE_GEOM(1, 4) MCLE_WAVE_GEOM(2, 4) MCLE_WAVE_GEOM(3, 4) MCLE_WAVE_GEOM(4, 4)
#undef MCLE_WAVE_GEOM
    return MCLE_E_UNSUPPORTED;
}

// the translation units (one per arithmetic, size and order mode: ten kernels each)
#define MCLE_MIMO_TDL_WAVE_ARGS                                                                                               \
    mcle_ctx *ctx, int nt, int nr, const MimoTdlParams &pp, int method, uint64_t seed, uint64_t first, uint64_t count,        \
        mcle_counters *d_counters, uint32_t *d_sym, uint32_t *d_bit
#define MCLE_MIMO_TDL_WAVE_TU(NAME, T_, N_, KT_)                                                                              \
    int NAME(MCLE_MIMO_TDL_WAVE_ARGS) {                                                                                       \
        return run_mimo_tdl_wave_size<T_, N_, KT_>(ctx, nt, nr, pp, method, seed, first, count, d_counters, d_sym, d_bit);    \
    }

}  // namespace mcle
