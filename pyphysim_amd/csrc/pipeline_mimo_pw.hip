// pipeline_mimo_pw.hip -- config 4's link (4 x 4 Blast + OFDM, complex128) with 1 / NW OF THE TIME SAMPLES PER WAVEFRONT, NW = 2, 4, 8
// (fft_size 512, 1024, 2048): the quarter-wave kernel of pipeline_mimo_qw.hip (NW = 4 there) as a template over the number of
// wavefronts, with the decode on the matrix cores (round 6).  Same link, same draw ledger (philox.hpp), same record kernel
// (k_mimo_filters_planar) and results contract as k_run_mimo_ofdm_planar<double, N, 4, 4, ...>, whose per-realization counts it
// reproduces (reference: apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:52-94, :394-466).
//
// Decomposition (pipeline_mimo_qw.hip has the long form): a workgroup of NW wavefronts is one realization, N = 256 NW.  Wavefront j
// owns the time samples n = NW m + j of ALL four antennas.  The first radix-NW DIF stage of the inverse transform,
//   x[NW m + j] = IDFT256_k' { conj(W_N)^(j k') sum_q e^(+2 pi i j q / NW) X[k' + 256 q] },
// is evaluated per wavefront for ITS j straight from the label bytes (NW table look-ups per element; recomputed, not exchanged);
// the 256-point transforms are two radix-16 register passes around one wave-private transposition; the channel is
// v_mfma_f64_4x4x4; the noise pair (2 p, 2 p + 1) is the same lane of wavefronts j and j ^ 1.  On receive the LAST radix-NW DIT stage
//   Y[k' + 256 q] = sum_j e^(-2 pi i j q / NW) W_N^(j k') Y_j[k']
// is the one exchange between the wavefronts (re planes, then im planes).
//
// What is new against pipeline_mimo_qw.hip: WHO reads the exchange.  There the thread that owns k' reads the sixteen (j, r) values
// and decodes four bins x four streams in its own registers: sixteen complex multiply-adds per bin on the VALU.  Here lane (r, g) of
// wavefront jw reads antenna r ONLY -- the NW partial transforms of its 16 / NW elements k' = g + 16 u, u in jw's share -- finishes
// the last stage for them (16 / NW butterflies of NW points) and holds Y_r at sixteen bins, while lanes (0..3, g) hold the SAME
// sixteen bins of the four receive antennas: the decode est_a = sum_r G[a][r] Y_r is the contraction over the four lanes of a column
// that the channel already is -- v_mfma_f64_4x4x4, lane (r, h) supplying G[h mod 4][r] -- and stream a's sixteen estimates land in
// lane (a, g).  Their labels are sixteen CONSECUTIVE bytes of that lane's label row ([q + NW u], u in jw's share): one 16-byte read.
// Decisions: the form fixed at compile time (walk_f64.hpp: walk_decide), four symbols at a time.
// LDS: NW planes of 8.5 KiB + tables -> 512: 27 KiB (five workgroups per CU fit, three wavefronts per SIMD = six workgroups of 128
// threads ... the register bound decides), 1024: 46 KiB (three workgroups per CU, as the quarter-wave kernel).
// Envelope: fft_size 256 NW, 4 x 4, full band, even cyclic prefix, a constellation with a certificate or the slicer.
#include "mimo_planar_common.hpp"
#include "walk_f64.hpp"

namespace mcle {

constexpr int kPwPlane = 4 * 272;                // doubles per wavefront plane (8 704 B): row stride 17 / 272, antenna stride 272
__host__ __device__ __forceinline__ int pw_slot(int a, int e) { return a * 272 + e + (e >> 4); }
// time index (within the wavefront's 256 samples) held by register c of lane group h after the second DIF pass
__host__ __device__ __forceinline__ int pw_mtime(int h, int c) { return (c & 3) * 64 + (c >> 2) * 16 + (h & 3) * 4 + (h >> 2); }
template <int NW> constexpr int pw_lab_stride() { return 16 * NW + 16; }     // bytes per (antenna, group) row: [q + NW u] + bank rotation

// Output J of the first radix-NW DIF stage for the sixteen elements k' = g + 16 u of one lane, from the label bytes of its row
// (byte q + NW u = the label of bin k' + 256 q): Z_u = conj(W_N^(16 J u)) sum_q e^(2 pi i J q / NW) X_q; the lane factor
// conj(W_N^(J g)) rides on the first pass's twiddles.
template <int NW, int J, bool STUB>
__device__ __forceinline__ void pw_first_stage(const unsigned char* lab_row, const double2* s_txtab, const double2* __restrict__ g_tw,
                                               double2 (&v)[16]) {
    const uint4* lab = reinterpret_cast<const uint4*>(lab_row);
    if constexpr (NW == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 L = lab[i];
            const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int u = 4 * i + uu;
                const uint32_t w = wds[uu];
                double2 X0, X1, X2, X3;
                if constexpr (STUB) {
                    X0 = X1 = X2 = X3 = mk<double>((double)w, 1.0);
                } else {
                    X0 = s_txtab[w & 0xFFu];
                    X1 = s_txtab[(w >> 8) & 0xFFu];
                    X2 = s_txtab[(w >> 16) & 0xFFu];
                    X3 = s_txtab[w >> 24];
                }
                const double2 A = (J & 1) ? csub(X0, X2) : cadd(X0, X2);
                const double2 B = (J & 1) ? csub(X1, X3) : cadd(X1, X3);
                double2 S;
                if constexpr (J == 0) S = cadd(A, B);
                else if constexpr (J == 1) S = mk<double>(A.x - B.y, A.y + B.x);      // A + i B
                else if constexpr (J == 2) S = csub(A, B);
                else S = mk<double>(A.x + B.y, A.y - B.x);                            // A - i B
                if constexpr (J == 0) v[u] = S;
                else if (u == 0) v[u] = S;
                else v[u] = cmulc(S, g_tw[16 * J * u]);                                // uniform address: a scalar load
            }
        }
    } else if constexpr (NW == 8) {
        // sum_q w^(J q) X_q, w = e^(2 pi i / 8), as (A_0 + i^J A_2) + w^J (A_1 + i^J A_3) with A_q0 = X_q0 + (-1)^J X_(q0 + 4)
        constexpr double kH = 0.70710678118654752440;
        auto rotJ = [](double2 z) {                               // i^J z
            if constexpr ((J & 3) == 0) return z;
            else if constexpr ((J & 3) == 1) return mk<double>(-z.y, z.x);
            else if constexpr ((J & 3) == 2) return mk<double>(-z.x, -z.y);
            else return mk<double>(z.y, -z.x);
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 L = lab[i];
            const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int u = 2 * i + uu;
                const uint32_t w0 = wds[2 * uu], w1 = wds[2 * uu + 1];
                double2 X[8];
                if constexpr (STUB) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) X[q] = mk<double>((double)(w0 + w1), 1.0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        X[q] = s_txtab[(w0 >> (8 * q)) & 0xFFu];
                        X[q + 4] = s_txtab[(w1 >> (8 * q)) & 0xFFu];
                    }
                }
                double2 A[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) A[q] = (J & 1) ? csub(X[q], X[q + 4]) : cadd(X[q], X[q + 4]);
                const double2 B0 = cadd(A[0], rotJ(A[2])), B1 = cadd(A[1], rotJ(A[3]));
                double2 S;
                if constexpr (J == 0) S = cadd(B0, B1);
                else if constexpr (J == 2) S = mk<double>(B0.x - B1.y, B0.y + B1.x);              // + i B1
                else if constexpr (J == 4) S = csub(B0, B1);
                else if constexpr (J == 6) S = mk<double>(B0.x + B1.y, B0.y - B1.x);              // - i B1
                else {
                    // w^J = (c + i s) / sqrt 2: J = 1: (1, 1), 3: (-1, 1), 5: (-1, -1), 7: (1, -1)
                    constexpr double c = (J == 1 || J == 7) ? kH : -kH, sn = (J == 1 || J == 3) ? kH : -kH;
                    S = mk<double>(B0.x + (c * B1.x - sn * B1.y), B0.y + (sn * B1.x + c * B1.y));
                }
                if constexpr (J == 0) v[u] = S;
                else if (u == 0) v[u] = S;
                else v[u] = cmulc(S, g_tw[16 * J * u]);
            }
        }
    } else {
        static_assert(NW == 2, "first stage");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint4 L = lab[i];
            const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                const int u = 8 * i + uu;
                const uint32_t w = wds[uu >> 1] >> (16 * (uu & 1));
                double2 X0, X1;
                if constexpr (STUB) {
                    X0 = X1 = mk<double>((double)(w & 0xFFFFu), 1.0);
                } else {
                    X0 = s_txtab[w & 0xFFu];
                    X1 = s_txtab[(w >> 8) & 0xFFu];
                }
                const double2 S = J ? csub(X0, X1) : cadd(X0, X1);
                if constexpr (J == 0) v[u] = S;
                else if (u == 0) v[u] = S;
                else v[u] = cmulc(S, g_tw[16 * J * u]);
            }
        }
    }
}

// ABL (MCLE_EXPERIMENTS builds only, option f64_variant: WRONG results by construction): 32 = no label draws / look-ups,
// 64 = no transmit passes, 128 = no noise draws, 256 = no channel products, 512 = no receive passes, 1024 = no decode
template <int NW, int DEC, int WPS, int ABL = 0>
__global__ __launch_bounds__(64 * NW, WPS) void k_run_mimo_ofdm_pw(MimoParams pp, ModemParams<double> mp, uint64_t seed, uint64_t first,
                                                                   uint64_t count, const double2* __restrict__ g_tw,
                                                                   const double2* __restrict__ g_recs, mcle_counters* counters,
                                                                   uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    using T = double;
    constexpr int N = 256 * NW, NT = 4, NR = 4, kRec = d64_rec<NT, NR>(), TB = 64 * NW, UU = 16 / NW;
    constexpr int kLabStride = pw_lab_stride<NW>();
    static_assert(NW == 2 || NW == 4 || NW == 8, "wavefronts per realization");
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
    T* s_R = reinterpret_cast<T*>(pw_smem);                                  // [NW wavefronts][kPwPlane]: scratch plane of wavefront j
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_R + NW * kPwPlane);           // [tab_len] constellation
    cx<T>* s_txtab = s_table + ((mp.M + 1) & ~1);                             // [tab_len] constellation x tx scale
    cx<T>* s_rec = s_txtab + ((mp.M + 1) & ~1);                               // [2][kRec + 1]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_rec + 2 * (kRec + 1));  // [2][16][2]
    constexpr int kBm = (kBmLdsDoubles + 1) & ~1;
    double* s_bm = reinterpret_cast<double*>(s_part + 64);                   // [kBm] Box-Muller tables
    unsigned char* s_lab = reinterpret_cast<unsigned char*>(s_bm + kBm);     // [4][16][kLabStride] labels (16-byte aligned)

    const int tid = threadIdx.x, lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);                 // this wavefront's time class n mod NW (scalar)
    const int pj = j & 1;
    const int cp = pp.cp;
    const int per_sym = N * NT;
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NT) / sqrt((double)(N + cp)));
    const uint32_t mask = (uint32_t)(mp.M - 1);
    for (int m = tid; m < mp.M; m += TB) {
        const cx<T> c = mp.g_table[m];
        s_table[m] = c;
        s_txtab[m] = cscale(c, tx_scale);
    }
    bm_tables_to_lds(s_bm, tid, TB);
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    T* s_mine = s_R + j * kPwPlane;
    uint2* s_words_mine = reinterpret_cast<uint2*>(s_mine);                // [16 slots][64 lanes] word pairs of MY samples
    uint2* s_words_peer = reinterpret_cast<uint2*>(s_R + (j ^ 1) * kPwPlane);  // ... of wavefront j ^ 1's
    uint64_t it = 0, rl_prev = 0;
    cx<T> rec_next = mk<T>(0, 0);
    if (tid < kRec && blockIdx.x < count) rec_next = g_recs[(uint64_t)blockIdx.x * kRec + tid];
    __syncthreads();
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x, ++it) {
        const Rng rng(seed, first + rl);
        const int buf = (int)(it & 1);
        if (tid < kRec) {                                                   // (first read after the next workgroup barrier)
            s_rec[buf * (kRec + 1) + tid] = rec_next;
            if (rl + gridDim.x < count) rec_next = g_recs[(rl + gridDim.x) * kRec + tid];
        }
        const cx<T>* s_H = s_rec + buf * (kRec + 1);                        // [NR][NT]
        const cx<T>* s_G = s_H + NT * NR;                                   // [NT][NR]
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            if (it > 0 || os > 0) __syncthreads();          // B5: the previous symbol's exchange planes and labels have been read
            // ---- S0a: this thread's DATA block: subcarriers d = 4 tid .. 4 tid + 3, four antennas each -> label bytes, laid out
            //      [antenna][group g = k' mod 16][q + NW u] for bin k = k' + 256 q, k' = g + 16 u (full band: k = d ^ (N / 2)) ----
            {
                const int t = opaque(tid);
                Words4 dw;
                if constexpr (ABL & 32) dw.w[0] = dw.w[1] = dw.w[2] = dw.w[3] = (uint32_t)t * 0x01010101u;
                else dw = rng.block(STREAM_DATA, (uint32_t)(((uint64_t)os * per_sym) >> 4) + (uint32_t)t);
                const int q = (t >> 6) ^ (NW / 2), u = (t & 63) >> 2;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t w = dw.w[s] & (mask * 0x01010101u);
                    const int g = 4 * (t & 3) + s;
                    unsigned char* dst = s_lab + g * kLabStride + NW * u + q;
#pragma unroll
                    for (int a = 0; a < 4; ++a) dst[a * 16 * kLabStride] = (unsigned char)(w >> (8 * a));
                }
            }
            // ---- S0b: the NOISE blocks of half of this lane's sixteen sample pairs: my two words stay, the partner's two go to
            //      wavefront j ^ 1 (same lane), both through the scratch planes (read back before the channel) ----
            {
                const int ln = opaque(lane);
                const int r = ln >> 4, h = ln & 15;
                // register c = 8 pj + cc holds sample time n = NW pw_mtime(h, c) + j, pw_mtime(h, c) = pw_mtime(h, cc) + 32 pj; the pair's
                // even sample has the flat index i0 = r row + os (N + cp) + cp + NW mtime + (j & ~1): block i0 / 2 = b0 + (NW / 2) pw_mtime(0, cc)
                const uint64_t i00 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + (j & ~1) + 32 * NW * pj + NW * (uint64_t)pw_mtime(h, 0);
                const uint32_t b0 = (uint32_t)(i00 >> 1);
                uint2* wm = s_words_mine + (8 * pj) * 64 + ln;
                uint2* wp = s_words_peer + (8 * pj) * 64 + ln;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    Words4 b;
                    if constexpr (ABL & 128) b.w[0] = b.w[1] = b.w[2] = b.w[3] = b0 + cc;
                    else b = rng.block(STREAM_NOISE, b0 + (uint32_t)(NW / 2) * (uint32_t)pw_mtime(0, cc));
                    const uint2 even = make_uint2(b.w[0], b.w[1]), odd = make_uint2(b.w[2], b.w[3]);
                    wm[cc * 64] = pj ? odd : even;
                    wp[cc * 64] = pj ? even : odd;
                }
            }
            __syncthreads();                                  // B1: labels and word pairs in place
            if (tid == 0 && os == 0 && it > 0) {              // every wave is past the previous realization: account it
                const unsigned* qq = s_part + (buf ^ 1) * 32;
                unsigned ts = 0, tb = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    ts += qq[2 * i];
                    tb += qq[2 * i + 1];
                }
                wg_account(totals, ts, tb, s_rec[(buf ^ 1) * (kRec + 1) + 2 * NT * NR].x != 0.0, rl_prev, sym_out, bit_out);
            }
            cx<T> v[16];
            // ---- S1: lane (a, g): the first-stage output of this wavefront's time class for k' = g + 16 u; compiled per class ----
            {
                const int ln = opaque(lane);
                const unsigned char* lab = s_lab + ln * kLabStride;                // (a * 16 + g) = lane
                if constexpr (NW == 4) {
                    switch (j) {
                        case 0: pw_first_stage<4, 0, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 1: pw_first_stage<4, 1, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 2: pw_first_stage<4, 2, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        default: pw_first_stage<4, 3, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                    }
                } else if constexpr (NW == 8) {
                    switch (j) {
                        case 0: pw_first_stage<8, 0, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 1: pw_first_stage<8, 1, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 2: pw_first_stage<8, 2, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 3: pw_first_stage<8, 3, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 4: pw_first_stage<8, 4, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 5: pw_first_stage<8, 5, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        case 6: pw_first_stage<8, 6, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                        default: pw_first_stage<8, 7, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                    }
                } else {
                    if (j == 0) pw_first_stage<2, 0, (ABL & 32) != 0>(lab, s_txtab, g_tw, v);
                    else pw_first_stage<2, 1, (ABL & 32) != 0>(lab, s_txtab, g_tw, v);
                }
            }
            // ---- transmit transform, pass 1 (DIF spans 64, 16 of the 256-point transforms; registers to registers) ----
            if constexpr (!(ABL & 64)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[g * (NW * m + j)];                       // W_256^(g m) x the lane factor W_N^(j g)
                    tw.a2[m - 1] = g_tw[4 * NW * g * m];
                }
                const cx<T> f0 = g_tw[g * j];
                r16_pass<T, true, false, 0, false, true, true>(nullptr, nullptr, 0, tw, nullptr, 0, v, v);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = cmulc(v[q], f0);             // row m' = 0 takes the lane factor by itself
            }
            // ---- the word pairs of my sixteen noise samples (before the scratch plane is reused) ----
            uint2 nw[16];
            {
                const int ln = opaque(lane);
#pragma unroll
                for (int c = 0; c < 16; ++c) nw[c] = s_words_mine[c * 64 + ln];
            }
            // ---- transposition (a, g | u) -> (a, h | c): element g + 16 u = 16 h + c, re plane then im plane ----
            {
                const int ln = opaque(lane);
                const int a = ln >> 4, g = ln & 15;
                const int wbase = pw_slot(a, g), rbase = pw_slot(a, 16 * g);       // element g + 16 u: wbase + 17 u; 16 g + c: rbase + c
                T xr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wbase + 17 * u] = v[u].x;
                walk_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) xr[c] = s_mine[rbase + c];
                walk_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wbase + 17 * u] = v[u].y;
                walk_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = mk<T>(xr[c], s_mine[rbase + c]);
            }
            // ---- pass 2 (spans 4, 1: sixteen consecutive elements, constant roots only) ----
            if constexpr (!(ABL & 64)) {
                R16Tw64<T> none;
                r16_pass<T, true, false, 0, false, true, true, false, true>(nullptr, nullptr, 0, none, nullptr, 0, v, v);
            }
            // ---- channel: R_r = sum_a H[r][a] T_a + noise on v_mfma_f64_4x4x4 (pipeline_mimo_qw.hip: the lane maps) ----
            {
                const int ln = opaque(lane);
                const cx<T> hA = s_H[(ln & 3) * NT + (ln >> 4)];                   // H[h mod 4][a]
                const T hre = hA.x, him = hA.y, nhim = -hA.y;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    cx<T> z;
                    if constexpr (ABL & 128) z = mk<T>((T)nw[c].x, sigma);
                    else z = cn_words(nw[c].x, nw[c].y, sigma, s_bm);
                    if constexpr (ABL & 256) {
                        v[c] = cadd(v[c], z);
                    } else {
                        T yr = __builtin_amdgcn_mfma_f64_4x4x4f64(hre, v[c].x, z.x, 0, 0, 0);
                        T yi = __builtin_amdgcn_mfma_f64_4x4x4f64(him, v[c].x, z.y, 0, 0, 0);
                        yr = __builtin_amdgcn_mfma_f64_4x4x4f64(nhim, v[c].y, yr, 0, 0, 0);
                        yi = __builtin_amdgcn_mfma_f64_4x4x4f64(hre, v[c].y, yi, 0, 0, 0);
                        v[c] = mk<T>(yr, yi);
                    }
                }
            }
            // ---- receive transform: pass 2' (DIT spans 1, 4), transposition back, pass 1' (spans 16, 64) ----
            if constexpr (!(ABL & 512)) {
                R16Tw64<T> none;
                r16_pass<T, false, true, 0, false, true, true, false, true>(nullptr, nullptr, 0, none, nullptr, 0, v, v);
            }
            {
                const int ln = opaque(lane);
                const int a = ln >> 4, g = ln & 15;
                const int wbase = pw_slot(a, g), rbase = pw_slot(a, 16 * g);
                T xr[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) s_mine[rbase + c] = v[c].x;
                walk_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) xr[u] = s_mine[wbase + 17 * u];
                walk_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) s_mine[rbase + c] = v[c].y;
                walk_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = mk<T>(xr[u], s_mine[wbase + 17 * u]);
            }
            if constexpr (!(ABL & 512)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[NW * g * m];
                    tw.a2[m - 1] = g_tw[4 * NW * g * m];
                }
                r16_pass<T, false, true, 0, false, true, true>(nullptr, nullptr, 0, tw, nullptr, 0, v, v);
            }
            // ---- the exchange: Y_j[k'] of receive antenna r (lane (r, g), register u: k' = g + 16 u) -> plane j at r 272 + k', re then
            //      im; lane (r, g) of wavefront jw reads the NW partial transforms of ITS k' = g + 16 (UU jw + uu), antenna r only ----
            T er[NW][UU], ei[NW][UU];
            {
                const int ln = opaque(lane);
                const int r = ln >> 4, g = ln & 15;
                const int wpos = r * 272 + g, rpos = r * 272 + g + 16 * UU * j;
                r16_wave_sync();                               // (my own reads of the transposition are done)
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wpos + 16 * u] = v[u].x;
                __syncthreads();                               // B2
#pragma unroll
                for (int jj = 0; jj < NW; ++jj)
#pragma unroll
                    for (int uu = 0; uu < UU; ++uu) er[jj][uu] = s_R[jj * kPwPlane + rpos + 16 * uu];
                __syncthreads();                               // B3
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wpos + 16 * u] = v[u].y;
                __syncthreads();                               // B4
#pragma unroll
                for (int jj = 0; jj < NW; ++jj)
#pragma unroll
                    for (int uu = 0; uu < UU; ++uu) ei[jj][uu] = s_R[jj * kPwPlane + rpos + 16 * uu];
            }
            // ---- last radix-NW stage for my UU elements (register q + NW uu = bin k' + 256 q), decode on the matrix cores, decisions ----
            if constexpr (!(ABL & 1024)) {
                const int ln = opaque(lane);
                const int g = ln & 15;
#pragma unroll
                for (int uu = 0; uu < UU; ++uu) {
                    const int kp = g + 16 * (UU * j + uu);
                    if constexpr (NW == 4) {
                        const cx<T> w1 = g_tw[kp], w2 = g_tw[2 * kp], w3 = g_tw[3 * kp];
                        const cx<T> u0 = mk<T>(er[0][uu], ei[0][uu]);
                        const cx<T> u1 = cmul(mk<T>(er[1][uu], ei[1][uu]), w1);
                        const cx<T> u2 = cmul(mk<T>(er[2][uu], ei[2][uu]), w2);
                        const cx<T> u3 = cmul(mk<T>(er[3][uu], ei[3][uu]), w3);
                        CxOps<T>::template bfly4<false>(u0, u1, u2, u3, v[4 * uu], v[4 * uu + 1], v[4 * uu + 2], v[4 * uu + 3]);
                    } else if constexpr (NW == 8) {
                        // Y_q = E_q + w'^q O_q, Y_(q + 4) = E_q - w'^q O_q, E / O = the 4-point forward transforms of the even / odd
                        // partial transforms, w' = e^(-2 pi i / 8)
                        cx<T> x[8];
                        x[0] = mk<T>(er[0][uu], ei[0][uu]);
#pragma unroll
                        for (int jj = 1; jj < 8; ++jj) x[jj] = cmul(mk<T>(er[jj][uu], ei[jj][uu]), g_tw[jj * kp]);
                        cx<T> E[4], O[4];
                        CxOps<T>::template bfly4<false>(x[0], x[2], x[4], x[6], E[0], E[1], E[2], E[3]);
                        CxOps<T>::template bfly4<false>(x[1], x[3], x[5], x[7], O[0], O[1], O[2], O[3]);
                        constexpr T kH = (T)0.70710678118654752440;
                        const cx<T> t0 = O[0];
                        const cx<T> t1 = mk<T>(kH * (O[1].x + O[1].y), kH * (O[1].y - O[1].x));      // (1 - i) / sqrt 2
                        const cx<T> t2 = mk<T>(O[2].y, -O[2].x);                                     // -i
                        const cx<T> t3 = mk<T>(kH * (O[3].y - O[3].x), -kH * (O[3].x + O[3].y));     // (-1 - i) / sqrt 2
                        v[8 * uu + 0] = cadd(E[0], t0); v[8 * uu + 4] = csub(E[0], t0);
                        v[8 * uu + 1] = cadd(E[1], t1); v[8 * uu + 5] = csub(E[1], t1);
                        v[8 * uu + 2] = cadd(E[2], t2); v[8 * uu + 6] = csub(E[2], t2);
                        v[8 * uu + 3] = cadd(E[3], t3); v[8 * uu + 7] = csub(E[3], t3);
                    } else {
                        const cx<T> u0 = mk<T>(er[0][uu], ei[0][uu]);
                        const cx<T> u1 = cmul(mk<T>(er[1][uu], ei[1][uu]), g_tw[kp]);
                        v[2 * uu] = cadd(u0, u1);
                        v[2 * uu + 1] = csub(u0, u1);
                    }
                }
                const uint4 L = *reinterpret_cast<const uint4*>(s_lab + ln * kLabStride + 16 * j);    // labels of my sixteen bins, stream a
                const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
                const cx<T> gA = s_G[(ln & 3) * NR + (ln >> 4)];                   // G[h mod 4][r]
                const T gre = gA.x, gim = gA.y, ngim = -gA.y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cx<T> e[4];
                    int tx[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int u = 4 * i + jj;
                        T xr = __builtin_amdgcn_mfma_f64_4x4x4f64(gre, v[u].x, 0.0, 0, 0, 0);
                        T xi = __builtin_amdgcn_mfma_f64_4x4x4f64(gim, v[u].x, 0.0, 0, 0, 0);
                        xr = __builtin_amdgcn_mfma_f64_4x4x4f64(ngim, v[u].y, xr, 0, 0, 0);
                        xi = __builtin_amdgcn_mfma_f64_4x4x4f64(gre, v[u].y, xi, 0, 0, 0);
                        e[jj] = mk<T>(xr, xi);
                        tx[jj] = (int)((wds[i] >> (8 * jj)) & 0xFFu);
                    }
                    walk_decide<DEC, 4>(mp, s_table, nullptr, e, tx, se, be);
                }
            } else {
                se += (unsigned)(er[0][0] + ei[NW - 1][UU - 1] == 0.5);
            }
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        if (lane == 0) {
            s_part[buf * 32 + 2 * j] = se;
            s_part[buf * 32 + 2 * j + 1] = be;
        }
        rl_prev = rl;
    }
    __syncthreads();
    if (tid == 0) {
        if (it > 0) {
            const int buf = (int)((it - 1) & 1);
            const unsigned* qq = s_part + buf * 32;
            unsigned ts = 0, tb = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                ts += qq[2 * i];
                tb += qq[2 * i + 1];
            }
            wg_account(totals, ts, tb, s_rec[buf * (kRec + 1) + 2 * NT * NR].x != 0.0, rl_prev, sym_out, bit_out);
        }
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym, (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
    }
}

template <int NW, int WPS, int ABL = 0>
static int launch_mimo_ofdm_pw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                               mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    using T = double;
    constexpr int N = 256 * NW, NT = 4, NR = 4, kRec = d64_rec<NT, NR>();
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, MCLE_F64, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const int dec = walk_dec_kind(ctx, mp);
    mp.grid.G = 0;
    const size_t tab_len = ((size_t)mp.M + 1) & ~(size_t)1;
    const size_t lds = (size_t)NW * kPwPlane * sizeof(T) + (2 * tab_len + 2 * (kRec + 1)) * sizeof(cx<T>) + 64 * sizeof(unsigned) +
                       (size_t)((kBmLdsDoubles + 1) & ~1) * sizeof(double) + (size_t)64 * pw_lab_stride<NW>();
    MCLE_REQUIRE(lds + 512 <= (size_t)160 * 1024, "part-wave MIMO-OFDM kernel: %zu B of LDS do not fit", lds);
    auto kern = k_run_mimo_ofdm_pw<NW, WDEC_SLICER, WPS, ABL>;
    switch (dec) {
        case WDEC_QAM_CERT: kern = k_run_mimo_ofdm_pw<NW, WDEC_QAM_CERT, WPS, ABL>; break;
        case WDEC_QUAD_CERT: kern = k_run_mimo_ofdm_pw<NW, WDEC_QUAD_CERT, WPS, ABL>; break;
        case WDEC_AXIS4_CERT: kern = k_run_mimo_ofdm_pw<NW, WDEC_AXIS4_CERT, WPS, ABL>; break;
        default: break;
    }
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    const int by_waves = WPS * 4 / NW;                         // wavefronts per SIMD x four SIMDs / wavefronts per workgroup
    if (per_cu > by_waves) per_cu = by_waves;
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 20;          // realizations per record kernel + link kernel pair (553 MB of records; 2^18: three more tails per bench step, -0.6 %)
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kRec * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL((k_mimo_filters_planar<T, N, NT, NR>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed,
                           first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        // (workgroups per resident slot: up to 32 at 1024 points -- 31.36 against 31.53 ms per 2^20 realizations at 16, 31.40 at 48)
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, n, 8, NW == 4 ? 32 : 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, ctx->stream, pp, mp, seed, first + off, n, (const cx<T>*)tw,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// 0 = launched; MCLE_E_UNSUPPORTED = outside the envelope (the caller stays on its other kernels)
int run_mimo_ofdm_pw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (!((cfg->fft_size == 512 || cfg->fft_size == 1024 || cfg->fft_size == 2048) && cfg->nt == 4 && cfg->nr == 4 && cfg->num_used == cfg->fft_size &&
          (cfg->cp_size & 1) == 0))
        return MCLE_E_UNSUPPORTED;
    if (ctx->M > 256) return MCLE_E_UNSUPPORTED;
    {
        ModemParams<double> mp = pipe_modem<double>(ctx, cfg->demod_method);
        if (walk_dec_kind(ctx, mp) == WDEC_GENERIC) return MCLE_E_UNSUPPORTED;     // no certificate: the planar kernel's candidate grid
    }
    const bool two = ctx->opt[MCLE_OPT_F64_THREADS] == 262 || ctx->opt[MCLE_OPT_F64_THREADS] == 264;
    if (cfg->fft_size == 2048)       // eight wavefronts = 512 threads: one workgroup per CU (86 KiB of LDS), two wavefronts per SIMD
        return launch_mimo_ofdm_pw<8, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    if (cfg->fft_size == 512) {
#ifdef MCLE_EXPERIMENTS
        switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
#define MCLE_PW_ABL(V_) case V_: return launch_mimo_ofdm_pw<2, 3, V_>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            MCLE_PW_ABL(32) MCLE_PW_ABL(64) MCLE_PW_ABL(128) MCLE_PW_ABL(256) MCLE_PW_ABL(512) MCLE_PW_ABL(1024) MCLE_PW_ABL(2016)
#undef MCLE_PW_ABL
            default: break;
        }
#endif
        return two ? launch_mimo_ofdm_pw<2, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
                   : launch_mimo_ofdm_pw<2, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    }
#ifdef MCLE_EXPERIMENTS
    switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
#define MCLE_PW_ABL(V_) case V_: return launch_mimo_ofdm_pw<4, 3, V_>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        MCLE_PW_ABL(32) MCLE_PW_ABL(64) MCLE_PW_ABL(128) MCLE_PW_ABL(256) MCLE_PW_ABL(512) MCLE_PW_ABL(1024) MCLE_PW_ABL(2016)
#undef MCLE_PW_ABL
        default: break;
    }
#endif
    return two ? launch_mimo_ofdm_pw<4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
               : launch_mimo_ofdm_pw<4, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
}

}  // namespace mcle
