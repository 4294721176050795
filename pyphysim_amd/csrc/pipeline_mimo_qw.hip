// pipeline_mimo_qw.hip -- config 4 at the benchmark geometry (4 x 4 Blast + 64-QAM + OFDM(1024), complex128) with ONE QUARTER OF
// THE TIME SAMPLES PER WAVEFRONT ("quarter-wave", round 6).  Same link, same draw ledger (philox.hpp), same record kernel
// (k_mimo_filters_planar) and the same results contract as k_run_mimo_ofdm_planar<double, 1024, 4, 4, 4, 2, 12>, whose
// per-realization counts it reproduces (reference: apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660,
// modulators/ofdm.py:394-466).
//
// Why: the planar kernel keeps a realization's 4 x 1024 complex128 samples in LDS between its stages -- 64 KiB of planes + 13 KiB
// of tables = two workgroups per CU, two wavefronts per SIMD, and 17 % of its time in barrier / first-LDS-round-trip waits that a
// third wavefront per SIMD would cover (profiles/r05/c4_f64_section_table.md).  64 KiB is what four antennas
// of 1024 points ARE, so the way below 53 KiB is to keep the samples in REGISTERS between the stages and use the LDS as a
// wave-private transposition scratch only.  That needs a decomposition in which a wavefront's data never leaves it:
//
//   wavefront j (of four) owns the time samples n = 4 m + j of ALL four antennas.  The 1024-point inverse transform splits by its
//   first radix-4 DIF stage into four 256-point transforms,  x[4 m + j] = IDFT256_k' { conj(W)^(j k') sum_q i^(j q) X[k' + 256 q] },
//   and a wavefront evaluates only ITS output j of that stage straight from the drawn symbol labels (four table look-ups per
//   element: the stage is recomputed per wavefront instead of exchanged -- 48 complex adds per lane).  The four 256-point
//   transforms of a wavefront are two radix-16 register passes (lane = (antenna, 16-point group), fft_r16.hpp: r16_pass from and to
//   registers) with ONE wave-local 16 x 16 transposition between them, through an 8 KiB plane of LDS, re then im.
//   After the second pass lane (a, h) holds sixteen samples of antenna a, and lanes (0..3, h) hold the same sixteen sample times:
//   the channel R = H T + noise is a reduce-scatter over those four lanes -- each lane forms its antenna's four partial products
//   and two swap-and-add steps (v_permlane32_swap, v_permlane16_swap: both directions of an exchange in one instruction) leave
//   receive antenna r with lane (r, h); no LDS.  The forward transform mirrors the two passes; its LAST radix-4 stage,
//   Y[k' + 256 q] = sum_j (-i)^(j q) W^(j k') Y_j[k'],  is the one exchange between the wavefronts: re planes, then im planes,
//   through the same 32 KiB, read by the thread that owns k' and decodes its four bins k' + 256 q in registers.
//   NOISE: a Philox block is the sample pair (2 p, 2 p + 1) -- the same lane of wavefronts j and j ^ 1.  Each of the two draws
//   half of the pair's blocks and hands the partner its two words through the partner's (then idle) scratch plane: the ledger is
//   unchanged (every block evaluated once per realization).
//
// LDS: 32 KiB scratch + tables (constellation x 2, Box-Muller, records, labels) = 46 KiB -> THREE workgroups per CU, three
// wavefronts per SIMD at a 168-register bound.  Five workgroup barriers per OFDM symbol, as before.
// Envelope: fft_size 1024, 4 x 4, full band (num_used = 1024), even cyclic prefix; anything else stays on the planar kernel.
#include "mimo_planar_common.hpp"

#ifndef MCLE_QW_ROLLED_DECODE
#define MCLE_QW_ROLLED_DECODE 0      // 1: the four bins of a thread in a rolled loop (A/B, profiles/r06/qw_ab.log)
#endif

namespace mcle {

constexpr int kQwLabStride = 80;                 // bytes per (antenna, group) row of labels: 64 + 16 (bank rotation, 16-byte aligned)
constexpr int kQwLabBytes = 4 * 16 * kQwLabStride;

// scratch slot of element e (0..255) of antenna a inside a wavefront's plane of doubles: one pad slot per sixteen elements (row
// stride 17, antenna stride 272 = 16 mod 32) -- conflict free for the 16-lane stores and the 32-lane loads of both transposition
// directions (tests/test_qw_layout.py replays them), and every access of a lane is base + compile-time offset: element g + 16 u
// sits at (272 a + g) + 17 u, element 16 h + c at (272 a + 17 h) + c (round 6, first edition: an XOR swizzle in 1024 slots -- an
// integer instruction per access, ~250 per wavefront and symbol)
constexpr int kQwPlane = 4 * 272;                // doubles per wavefront plane (8 704 B)
__host__ __device__ __forceinline__ int qw_slot(int a, int e) { return a * 272 + e + (e >> 4); }
// time index (within the wavefront's 256 samples) held by register c of lane group h after the second DIF pass
__host__ __device__ __forceinline__ int qw_mtime(int h, int c) { return (c & 3) * 64 + (c >> 2) * 16 + (h & 3) * 4 + (h >> 2); }

// ordering of a wavefront's OWN LDS traffic (the transpositions through its private plane): the DS queue executes a wavefront's
// instructions in order, so a store followed by a load of another lane's slot needs no wait -- only the compiler must keep the
// program order (fft_r16.hpp's r16_wave_sync fences at workgroup scope, i.e. s_waitcnt lgkmcnt(0): a full LDS round trip per phase)
__device__ __forceinline__ void qw_wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void swap16_pair(double a, double b, double& x, double& y) {
    // rows of 16 lanes: x = {even rows: own a, odd rows: b of lane - 16}, y = {even rows: a of lane + 16, odd rows: own b}
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// Output J of the first radix-4 DIF stage for the sixteen elements k' = g + 16 u of one lane, from the label bytes (word u of the
// lane's row = the labels of the bins k' + 256 q, q = 0..3): Z_u = conj(W^(16 J u)) (X_0 + i^J X_1 + i^(2J) X_2 + i^(3J) X_3)
template <int J, bool STUB>
__device__ __forceinline__ void qw_first_stage(const unsigned char* lab_row, const double2* s_txtab, const double2* __restrict__ g_tw,
                                               double2 (&v)[16]) {
    const uint4* lab = reinterpret_cast<const uint4*>(lab_row);
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
}

// ABL (MCLE_EXPERIMENTS builds only, option f64_variant: WRONG results by construction, the section table's ablations):
// 32 = no label draws / look-ups, 64 = no transmit passes, 128 = no noise draws, 256 = no channel products, 512 = no receive passes,
// 1024 = no decode
template <int WPS, int ABL = 0>
__global__ __launch_bounds__(256, WPS) void k_run_mimo_ofdm_qw(MimoParams pp, ModemParams<double> mp, uint64_t seed, uint64_t first,
                                                               uint64_t count, const double2* __restrict__ g_tw,
                                                               const double2* __restrict__ g_recs, mcle_counters* counters,
                                                               uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    using T = double;
    constexpr int N = 1024, NT = 4, NR = 4, kRec = d64_rec<NT, NR>(), NW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* s_R = reinterpret_cast<T*>(smem);                                   // [4 wavefronts][kQwPlane]: scratch plane of wavefront j
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_R + 4 * kQwPlane);             // [tab_len] constellation
    cx<T>* s_txtab = s_table + ((mp.M + 1) & ~1);                          // [tab_len] constellation x tx scale
    cx<T>* s_rec = s_txtab + ((mp.M + 1) & ~1);                            // [2][kRec + 1]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_rec + 2 * (kRec + 1));  // [2][16][2]
    constexpr int kBm = (kBmLdsDoubles + 1) & ~1;
    double* s_bm = reinterpret_cast<double*>(s_part + 64);                 // [kBm] Box-Muller tables
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_bm + kBm);
    unsigned char* s_lab = reinterpret_cast<unsigned char*>(                                   // [4][16][80] labels, 16-byte aligned
        (reinterpret_cast<uintptr_t>(s_grid + mp.grid.G * mp.grid.G) + 15) & ~(uintptr_t)15);

    const int tid = threadIdx.x, lane = tid & 63;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);               // this wavefront's time class n mod 4 (scalar)
    const int pj = j & 1;
    const int cp = pp.cp;
    const int per_sym = N * NT;
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NT) / sqrt((double)(N + cp)));
    const uint32_t mask = (uint32_t)(mp.M - 1);
    for (int m = tid; m < mp.M; m += 256) {
        const cx<T> c = mp.g_table[m];
        s_table[m] = c;
        s_txtab[m] = cscale(c, tx_scale);
    }
    load_grid(mp, s_grid);
    bm_tables_to_lds(s_bm, tid, 256);
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    T* s_mine = s_R + j * kQwPlane;
    uint2* s_words_mine = reinterpret_cast<uint2*>(s_mine);                // [16 slots][64 lanes] word pairs of MY samples
    uint2* s_words_peer = reinterpret_cast<uint2*>(s_R + (j ^ 1) * kQwPlane);  // ... of wavefront j ^ 1's
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
            //      [antenna][group g = k' mod 16][q + 4 u] for bin k = k' + 256 q, k' = g + 16 u (full band: k = d ^ 512) ----
            {
                const int t = opaque(tid);
                Words4 dw;
                if constexpr (ABL & 32) dw.w[0] = dw.w[1] = dw.w[2] = dw.w[3] = (uint32_t)t * 0x01010101u;
                else dw = rng.block(STREAM_DATA, (uint32_t)(((uint64_t)os * per_sym) >> 4) + (uint32_t)t);
                const int q = (t >> 6) ^ 2, u = (t & 63) >> 2;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t w = dw.w[s] & (mask * 0x01010101u);
                    const int g = 4 * (t & 3) + s;
                    unsigned char* dst = s_lab + g * kQwLabStride + 4 * u + q;
#pragma unroll
                    for (int a = 0; a < 4; ++a) dst[a * 16 * kQwLabStride] = (unsigned char)(w >> (8 * a));
                }
            }
            // ---- S0b: the NOISE blocks of half of this lane's sixteen sample pairs: my two words stay, the partner's two go to
            //      wavefront j ^ 1 (same lane), both through the scratch planes (read back before the first transposition) ----
            {
                const int ln = opaque(lane);
                const int r = ln >> 4, h = ln & 15;
                // sample time of slot c = 8 pj + cc: qw_mtime(h, c) = qw_mtime(h, cc) + 32 pj, and the pair's even sample has the flat
                // index i0 = r row + os (N + cp) + cp + 4 mtime + (j & 2) (even): block i0 / 2 = b0 + 2 qw_mtime(0, cc)
                const uint64_t i00 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + (j & 2) + 128 * pj + 4 * (uint64_t)qw_mtime(h, 0);
                const uint32_t b0 = (uint32_t)(i00 >> 1);
                uint2* wm = s_words_mine + (8 * pj) * 64 + ln;
                uint2* wp = s_words_peer + (8 * pj) * 64 + ln;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    Words4 b;
                    if constexpr (ABL & 128) b.w[0] = b.w[1] = b.w[2] = b.w[3] = b0 + cc;
                    else b = rng.block(STREAM_NOISE, b0 + 2u * (uint32_t)qw_mtime(0, cc));
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
            // ---- S1: lane (a, g): Z[g + 16 u] = conj(W^(16 j u)) sum_q i^(j q) X_a[g + 16 u + 256 q], u = 0..15 (the lane factor
            //      conj(W^(j g)) rides on the first pass's twiddles); compiled per time class: the rotations i^(j q) are adds with
            //      signs and the sixteen constants W^(16 j u) scalar loads at immediate offsets ----
            {
                const int ln = opaque(lane);
                const unsigned char* lab = s_lab + ln * kQwLabStride;              // (a * 16 + g) = lane
                switch (j) {
                    case 0: qw_first_stage<0, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                    case 1: qw_first_stage<1, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                    case 2: qw_first_stage<2, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                    default: qw_first_stage<3, (ABL & 32) != 0>(lab, s_txtab, g_tw, v); break;
                }
            }
            // ---- transmit transform, pass 1 (DIF spans 64, 16 of the 256-point transforms; registers to registers) ----
            if constexpr (!(ABL & 64)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[g * (4 * m + j)];                        // W^(4 g m) x the lane factor W^(j g)
                    tw.a2[m - 1] = g_tw[16 * g * m];
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
                const int wbase = qw_slot(a, g), rbase = qw_slot(a, 16 * g);       // element g + 16 u: wbase + 17 u; 16 g + c: rbase + c
                T xr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wbase + 17 * u] = v[u].x;
                qw_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) xr[c] = s_mine[rbase + c];
                qw_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[wbase + 17 * u] = v[u].y;
                qw_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = mk<T>(xr[c], s_mine[rbase + c]);
            }
            // ---- pass 2 (spans 4, 1: sixteen consecutive elements, constant roots only) ----
            if constexpr (!(ABL & 64)) {
                R16Tw64<T> none;
                r16_pass<T, true, false, 0, false, true, true, false, true>(nullptr, nullptr, 0, none, nullptr, 0, v, v);
            }
            // ---- channel: lane (a, h) holds T_a at sixteen sample times, and R_r = sum_a H[r][a] T_a + noise has to end in lane (r, h).
            //      That IS v_mfma_f64_4x4x4 (four independent 4 x 4 x 4 products per instruction; lane maps measured in round 3,
            //      profiles/r03/f64_rates.txt: A_b[i][k] <- lane 4 b + i + 16 k, B_b[k][j] <- lane 4 b + j + 16 k, D_b[i][j] -> lane
            //      4 b + j + 16 i): with k = the transmit antenna (the lane's row), j + 4 b = h and i = the receive antenna, B is the
            //      lane's own sample, D lands in the receive antenna's row, and A_b[i][k] = H[i][k] for every block -- lane (a, h)
            //      supplies H[h mod 4][a].  Four instructions per sample time (re / im of H x re / im of T; the noise sample is the
            //      accumulator's start) where the VALU form was 16 multiply-adds + 6 adds + 12 lane swaps: the batched
            //      Nt x Nr x Ns contraction on the matrix cores (BASELINE.json north_star). ----
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
                const int wbase = qw_slot(a, g), rbase = qw_slot(a, 16 * g);
                T xr[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) s_mine[rbase + c] = v[c].x;
                qw_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) xr[u] = s_mine[wbase + 17 * u];
                qw_wave_order();
#pragma unroll
                for (int c = 0; c < 16; ++c) s_mine[rbase + c] = v[c].y;
                qw_wave_order();
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = mk<T>(xr[u], s_mine[wbase + 17 * u]);
            }
            if constexpr (!(ABL & 512)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[4 * g * m];
                    tw.a2[m - 1] = g_tw[16 * g * m];
                }
                r16_pass<T, false, true, 0, false, true, true>(nullptr, nullptr, 0, tw, nullptr, 0, v, v);
            }
            // ---- the exchange: Y_j[k'] of receive antenna r (lane (r, g), register u: k' = g + 16 u) -> plane j, re then im; the
            //      thread that owns k' = tid reads the sixteen (j, r) values of each ----
            T er[4][4], ei[4][4];
            {
                const int ln = opaque(lane);
                const int r = ln >> 4, g = ln & 15;
                r16_wave_sync();                               // (my own reads of the transposition are done)
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[r * 256 + g + 16 * u] = v[u].x;
                __syncthreads();                               // B2
                const int t = opaque(tid);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) er[jj][rr] = s_R[jj * kQwPlane + rr * 256 + t];
                __syncthreads();                               // B3
#pragma unroll
                for (int u = 0; u < 16; ++u) s_mine[r * 256 + g + 16 * u] = v[u].y;
                __syncthreads();                               // B4
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) ei[jj][rr] = s_R[jj * kQwPlane + rr * 256 + t];
            }
            // ---- last radix-4 stage (DIT span 256) + Blast decode of the four bins k' + 256 q, demodulate, count ----
            if constexpr (!(ABL & 1024)) {
                const int t = opaque(tid);
                const cx<T> w1 = g_tw[t], w2 = g_tw[2 * t], w3 = g_tw[3 * t];
                cx<T> Y[4][NR];                                 // [q][r]
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    const cx<T> u0 = mk<T>(er[0][rr], ei[0][rr]);
                    const cx<T> u1 = cmul(mk<T>(er[1][rr], ei[1][rr]), w1);
                    const cx<T> u2 = cmul(mk<T>(er[2][rr], ei[2][rr]), w2);
                    const cx<T> u3 = cmul(mk<T>(er[3][rr], ei[3][rr]), w3);
                    CxOps<T>::template bfly4<false>(u0, u1, u2, u3, Y[0][rr], Y[1][rr], Y[2][rr], Y[3][rr]);
                }
                uint32_t labw[NT];
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    labw[a] = *reinterpret_cast<const uint32_t*>(s_lab + (a * 16 + (t & 15)) * kQwLabStride + 4 * (t >> 4));
                // Four bins per thread, straight-line: the filter, then the slicer or the margin certificates of the four streams.  A
                // symbol its certificate does not vouch for (one chance in ~1e8 under the QAM certificate) goes through the plain
                // SWEEP over the table -- a dozen instructions of rolled loop per stream, not the inlined lockstep candidate-grid search
                // of the planar kernels: the first edition of this kernel inlined that search four times, 114 KB of decode in a 150 KB
                // kernel against an instruction cache of 64 KB per two CUs (and a rolled decode loop around one copy of it spilled 63
                // registers at the 168-register bound).  The launcher sends constellations WITHOUT a certificate to the planar kernel.
#if MCLE_QW_ROLLED_DECODE
#pragma unroll 1
#else
#pragma unroll
#endif
                for (int q = 0; q < 4; ++q) {
                    cx<T> est[NT];
                    int dec[NT];
                    constexpr bool kRolled = MCLE_QW_ROLLED_DECODE != 0;
                    const int qi = kRolled ? 0 : q;                          // rolled: the rows of Y move up by one after every bin
#pragma unroll
                    for (int a = 0; a < NT; ++a) {
                        est[a] = mk<T>(0, 0);
#pragma unroll
                        for (int rr = 0; rr < NR; ++rr) est[a] = cfma4(s_G[a * NR + rr], Y[qi][rr], est[a]);
                    }
                    if (mp.method == MCLE_DEMOD_QAM_SLICER) {
#pragma unroll
                        for (int a = 0; a < NT; ++a) dec[a] = demod_qam_slicer<T>(est[a], mp.qam_scale, mp.qam_L, mp.half_bits);
                    } else {
                        demod_multi_cert(mp, est, dec, [&](int (&d_)[NT]) {
#pragma unroll 1
                            for (int a = 0; a < NT; ++a) d_[a] = demod_mindist<T>(s_table, mp.M, est[a]);
                        });
                    }
#pragma unroll
                    for (int a = 0; a < NT; ++a) {
                        const unsigned x = (labw[a] & 0xFFu) ^ (unsigned)dec[a];
                        se += (x != 0u);
                        be += __popc(x);
                        labw[a] >>= 8;
                    }
                    if constexpr (kRolled) {
#pragma unroll
                        for (int rr = 0; rr < NR; ++rr) {
                            Y[0][rr] = Y[1][rr];
                            Y[1][rr] = Y[2][rr];
                            Y[2][rr] = Y[3][rr];
                        }
                    }
                }
            } else {
                se += (unsigned)(er[0][0] + ei[3][3] == 0.5);
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

template <int WPS, int ABL = 0>
static int launch_mimo_ofdm_qw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                               mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    using T = double;
    constexpr int N = 1024, NT = 4, NR = 4, kRec = d64_rec<NT, NR>();
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, MCLE_F64, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const size_t tab_len = ((size_t)mp.M + 1) & ~(size_t)1;
    const size_t lds = (size_t)4 * kQwPlane * sizeof(T) + (2 * tab_len + 2 * (kRec + 1)) * sizeof(cx<T>) + 64 * sizeof(unsigned) +
                       (size_t)((kBmLdsDoubles + 1) & ~1) * sizeof(double) + (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       kQwLabBytes + 16;
    MCLE_REQUIRE(lds + 512 <= (size_t)160 * 1024, "quarter-wave MIMO-OFDM kernel: %zu B of LDS do not fit", lds);
    auto kern = k_run_mimo_ofdm_qw<WPS, ABL>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > WPS) per_cu = WPS;                            // 256 threads = one wavefront per SIMD and workgroup
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 18;
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kRec * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL((k_mimo_filters_planar<T, N, NT, NR>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed,
                           first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, n, 8, 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, pp, mp, seed, first + off, n, (const cx<T>*)tw,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// 0 = launched; MCLE_E_UNSUPPORTED = outside the envelope (the caller stays on the planar kernel)
int run_mimo_ofdm_qw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (!(cfg->fft_size == 1024 && cfg->nt == 4 && cfg->nr == 4 && cfg->num_used == 1024 && (cfg->cp_size & 1) == 0))
        return MCLE_E_UNSUPPORTED;
    if (ctx->M > 256) return MCLE_E_UNSUPPORTED;
    // min-distance decisions through a certificate (square QAM, QPSK) or the slicer: the sweep behind the certificate is the rare path here
    if (cfg->demod_method != MCLE_DEMOD_QAM_SLICER && modem_cert(ctx, cfg->demod_method) == 0) return MCLE_E_UNSUPPORTED;
#ifdef MCLE_EXPERIMENTS
    switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
#define MCLE_QW_ABL(V_) case V_: return launch_mimo_ofdm_qw<3, V_>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        MCLE_QW_ABL(32) MCLE_QW_ABL(64) MCLE_QW_ABL(128) MCLE_QW_ABL(256) MCLE_QW_ABL(512) MCLE_QW_ABL(1024) MCLE_QW_ABL(2016) MCLE_QW_ABL(384)
#undef MCLE_QW_ABL
        default: break;
    }
#endif
    if (ctx->opt[MCLE_OPT_F64_THREADS] == 262) return launch_mimo_ofdm_qw<2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    return launch_mimo_ofdm_qw<3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
}

}  // namespace mcle
