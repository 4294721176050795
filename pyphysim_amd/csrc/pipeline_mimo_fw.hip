// pipeline_mimo_fw.hip -- config 4's link at fft_size 256 (4 x 4 Blast + OFDM(256), complex128) with ONE REALIZATION PER WAVEFRONT
// ("full-wave", round 6).  Same link, same draw ledger (philox.hpp), same record kernel (k_mimo_filters_planar) and the same
// results contract as k_run_mimo_ofdm_planar<double, 256, 4, 4, ...>, whose per-realization counts it reproduces (reference:
// apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:394-466; any fft_size: modulators/ofdm.py:52-94).
//
// Why: off the benchmark size the planar family ran radix-4 stages through LDS planes with a workgroup
// barrier per stage group -- 0.68 of the quarter-wave kernel's per-subcarrier rate at 256 points.  The quarter-wave kernel
// (pipeline_mimo_qw.hip) already holds a 256-point transform of all four antennas in ONE wavefront -- lane = (antenna, 16-point
// group), two radix-16 register passes with one wave-private 16 x 16 transposition between them -- and needs the other three
// wavefronts only for the radix-4 stage that makes 1024 out of 4 x 256.  At fft_size 256 that stage does not exist, so the whole
// realization is one wavefront's: NO workgroup barrier, no exchange planes, and
//   * labels: lane t draws DATA block t (four subcarriers x four antennas); the bytes go through a 1 KiB array [antenna][k mod 16]
//     [k div 16] and lane (a, g) reads its sixteen labels X_a[g + 16 u] as ONE 16-byte word, which it keeps for the decode;
//   * transmit: sixteen table look-ups, pass 1 (spans 64, 16), transposition through the wavefront's 8.5 KiB plane (re, then im),
//     pass 2 (constant roots only) -- fft_r16.hpp: r16_pass from and to registers, as in the quarter-wave kernel;
//   * noise: register c of lane (r, h) is sample time n = qw_mtime(h, c), so the pair (2 p, 2 p + 1) of a Philox NOISE block is the
//     lane pair (l, l ^ 4): each evaluates the blocks of eight of the sixteen registers and hands the partner its two words
//     through the (then idle) plane;
//   * channel R = H T + noise on v_mfma_f64_4x4x4 (four instructions per sample time, the lane maps of the quarter-wave kernel);
//   * receive: the mirror image; lane (r, g) ends with Y_r[g + 16 u] -- and the DECODE est_a = sum_r G[a][r] Y_r is the same
//     contraction over the four lanes of a column: v_mfma_f64_4x4x4 again, lane (r, h) supplying G[h mod 4][r], and stream a's
//     sixteen estimates land in lane (a, g) -- next to the sixteen labels that lane read at the top.  Decisions: the form fixed at
//     compile time (slicer, QAM margin certificate, quadrant certificate; walk_f64.hpp: walk_decide), four symbols at a time.
// A workgroup is four independent wavefronts that share the tables (constellation x 2, Box-Muller): 46 KiB of LDS -> three
// workgroups per CU, three wavefronts per SIMD at a 168-register bound.
// 2 x 2: TWO realizations per wavefront (lane = (realization, antenna, group)); the 4 x 4 x 4 contractions take a block-diagonal A.
// complex64 (k_run_mimo_ofdm_fw<float, ...>): the same kernel with the two contractions as reduce-scatters on the VALU (fw_contract:
// products per lane, v_permlane32_swap / v_permlane16_swap and adds) -- the f32 MFMA forms have K = 1 and cannot contract across lanes.
// Envelope: fft_size 256, 4 x 4 or 2 x 2, full band (num_used = 256), even cyclic prefix, a constellation with a certificate or the
// slicer; anything else stays on the planar kernel.
#include "mimo_planar_common.hpp"
#include "walk_f64.hpp"

namespace mcle {

constexpr int kFwPlane = 4 * 272;                // doubles per wavefront plane (8 704 B): row stride 17, antenna stride 272
constexpr int kFwLabBytes = 64 * 16;             // [antenna][k mod 16][k div 16] label bytes of one wavefront
__host__ __device__ __forceinline__ int fw_slot(int a, int e) { return a * 272 + e + (e >> 4); }
// sample time held by register c of lane group h after the second DIF pass (= pipeline_mimo_qw.hip: qw_mtime)
__host__ __device__ __forceinline__ int fw_mtime(int h, int c) { return (c & 3) * 64 + (c >> 2) * 16 + (h & 3) * 4 + (h >> 2); }

// ABL (MCLE_EXPERIMENTS builds only, option f64_variant: WRONG results by construction): 32 = no label draws / look-ups,
// 64 = no transmit passes, 128 = no noise draws, 256 = no channel products, 512 = no receive passes, 1024 = no decode
// NA = antennas per side (Nt = Nr = NA): 4 = one realization per wavefront; 2 = TWO realizations per wavefront -- lane = (realization,
// antenna, 16-point group), the contractions on the same instruction with a block-diagonal A operand (lane (row, col) supplies
// H[col mod 2][row mod 2] of ITS realization where row div 2 = col div 2 mod 2, and 0 elsewhere), per-lane Philox counters.
// lanes l and l ^ 16 exchange (v_permlane16_swap_b32): x = {even rows: own a, odd rows: the partner's b}, y = {even rows: the partner's a,
// odd rows: own b}
__device__ __forceinline__ void fw_swap16_pair(float a, float b, float& x, float& y) {
    const auto v = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    x = __uint_as_float(v[0]);
    y = __uint_as_float(v[1]);
}
// complex64 (the last day of round 6): the contraction over the four lanes of a column as a reduce-scatter on the VALU -- the f32 MFMA
// forms have K = 1 (no contraction across lanes).  Lane (row k, col) holds x_k and the row-k COLUMN M[0..NA-1][k] of the matrix; every lane
// forms its NA products, and one (NA = 2) or two (NA = 4) swap-and-add steps leave sum_k M[i][k] x_k + acc_i in lane (row i, col):
// rows k and k ^ 2 (v_permlane32_swap), then k and k ^ 1 (v_permlane16_swap).  NA = 2: two realizations per wavefront, rows (rz, a):
// only the second step, which stays inside a realization.
template <int NA>
__device__ __forceinline__ float2 fw_contract(const float2 (&Mc)[NA], float2 x, float2 acc) {
    float2 p[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) p[i] = cmul(Mc[i], x);
    float x0, y0, x1, y1;
    if constexpr (NA == 4) {
        float2 s02, s13;
        swap32_pair(p[0].x, p[2].x, x0, y0);
        swap32_pair(p[0].y, p[2].y, x1, y1);
        s02 = make_float2(x0 + y0, x1 + y1);
        swap32_pair(p[1].x, p[3].x, x0, y0);
        swap32_pair(p[1].y, p[3].y, x1, y1);
        s13 = make_float2(x0 + y0, x1 + y1);
        fw_swap16_pair(s02.x, s13.x, x0, y0);
        fw_swap16_pair(s02.y, s13.y, x1, y1);
    } else {
        fw_swap16_pair(p[0].x, p[1].x, x0, y0);
        fw_swap16_pair(p[0].y, p[1].y, x1, y1);
    }
    return make_float2((x0 + y0) + acc.x, (x1 + y1) + acc.y);
}

template <typename T> constexpr int fw_plane_bytes() { return sizeof(T) == 8 ? kFwPlane * 8 : 8192; }   // (>= the 8 KiB of the word-pair hand-over)

template <typename T, int NA, int DEC, int WPS, int ABL = 0>
__global__ __launch_bounds__(256, WPS) void k_run_mimo_ofdm_fw(MimoParams pp, ModemParams<T> mp, uint64_t seed, uint64_t first,
                                                               uint64_t count, const cx<T>* __restrict__ g_tw,
                                                               const cx<T>* __restrict__ g_recs, mcle_counters* counters,
                                                               uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    constexpr int N = 256, NT = NA, NR = NA, kRec = d64_rec<NT, NR>(), RZ = 4 / NA, LPR = 64 / RZ;     // realizations / lanes per realization
    static_assert(NA == 2 || NA == 4, "geometry");
    extern __shared__ __attribute__((aligned(16))) char fw_smem[];
    constexpr int kPlaneT = fw_plane_bytes<T>() / (int)sizeof(T);            // scalars per wavefront plane
    T* s_R = reinterpret_cast<T*>(fw_smem);                                  // [4 wavefronts][kPlaneT]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_R + 4 * kPlaneT);             // [tab_len] constellation
    cx<T>* s_txtab = s_table + ((mp.M + 1) & ~1);                             // [tab_len] constellation x tx scale
    cx<T>* s_rec = s_txtab + ((mp.M + 1) & ~1);                               // [4 wavefronts][RZ][kRec + 1]
    constexpr int kBm = sizeof(T) == 8 ? ((kBmLdsDoubles + 1) & ~1) : 0;
    double* s_bm = reinterpret_cast<double*>(reinterpret_cast<char*>(s_rec + 4 * RZ * (kRec + 1)) + (sizeof(T) == 4 ? 8 : 0));    // [kBm] Box-Muller tables (complex128)
    unsigned char* s_lab = reinterpret_cast<unsigned char*>(s_bm + kBm);     // [4][kFwLabBytes] (16-byte aligned: everything before is)
    __shared__ WgTotals totals[4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, tid, 256);
    if (lane == 0) wg_zero(totals[w]);
    __syncthreads();                                                         // the only workgroup barrier: the shared tables

    T* s_mine = s_R + w * kPlaneT;
    uint2* s_words = reinterpret_cast<uint2*>(s_mine);                      // [16 registers][64 lanes] word pairs (8 KiB of the plane)
    unsigned char* lab_mine = s_lab + w * kFwLabBytes;
    cx<T>* rec_mine = s_rec + w * RZ * (kRec + 1);
    const uint64_t stride = (uint64_t)gridDim.x * 4 * RZ;
    const int rz = lane / LPR, lr = lane - rz * LPR;                        // this lane's realization of the wavefront's RZ, lane within it
    cx<T> rec_next = mk<T>(0, 0);
    {
        const uint64_t b0 = ((uint64_t)blockIdx.x * 4 + w) * RZ;
        const uint64_t r0 = b0 + rz < count ? b0 + rz : b0;
        if (lr < kRec && r0 < count) rec_next = g_recs[r0 * kRec + lr];
    }
    for (uint64_t rl0 = ((uint64_t)blockIdx.x * 4 + w) * RZ; rl0 < count; rl0 += stride) {
        // (NA = 2: past the end the wavefront's second half repeats the first half's realization -- finite values in the block-diagonal
        //  contractions' zero blocks -- and is not accounted)
        const uint64_t rl = rl0 + (uint64_t)rz < count ? rl0 + (uint64_t)rz : rl0;
        const Rng rng(seed, first + rl);
        walk_wave_order();                                                   // (the previous realization's reads of the record are done)
        if (lr < kRec) {
            rec_mine[rz * (kRec + 1) + lr] = rec_next;
            const uint64_t nb = rl0 + stride, nx = nb + rz < count ? nb + rz : nb;
            if (nx < count) rec_next = g_recs[nx * kRec + lr];
        }
        walk_wave_order();
        const int ln0 = opaque(lane);
        // A operands of the two contractions: lane (row, col) supplies M[col mod 4][row]; with two realizations per wavefront the
        // 4 x 4 matrix is block diagonal (row = (realization, antenna), col mod 4 = (realization', antenna'))
        cx<T> hA, gA;
        {
            const int col = ln0 & 3, rowi = ln0 >> 4;
            const int rzc = col / NA, rr = col % NA, rzr = rowi / NA, aa = rowi % NA;
            const cx<T>* rc = rec_mine + rzr * (kRec + 1);
            hA = rc[rr * NT + aa];                                           // H[rr][aa]
            gA = rc[NT * NR + rr * NR + aa];                                 // G[rr][aa]
            if (NA < 4 && rzc != rzr) hA = gA = mk<T>(0, 0);
        }
        // complex64: the lane's COLUMNS of H and G for the VALU contraction (row k = this lane's antenna: H[0..NR-1][k], G[0..NT-1][k])
        [[maybe_unused]] cx<T> Hc[NA], Gc[NA];
        if constexpr (sizeof(T) == 4) {
            const int rowi = ln0 >> 4, rzr = rowi / NA, aa = rowi % NA;
            const cx<T>* rc = rec_mine + rzr * (kRec + 1);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                Hc[i] = rc[i * NT + aa];
                Gc[i] = rc[NT * NR + i * NR + aa];
            }
        }
        const bool skipped = rec_mine[rz * (kRec + 1) + 2 * NT * NR].x != (T)0;
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            // ---- labels: DATA block `lane` of the symbol = subcarriers d = 4 lane .. 4 lane + 3, four antennas each (full band:
            //      bin k = d ^ 128) -> byte [antenna][k mod 16][k div 16]; lane (a, g) then reads its sixteen as one word ----
            uint4 L;
            {
                const int t = opaque(lane);
                Words4 dw;
                if constexpr (ABL & 32) dw.w[0] = dw.w[1] = dw.w[2] = dw.w[3] = (uint32_t)t * 0x01010101u;
                else dw = rng.block(STREAM_DATA, (uint32_t)(((uint64_t)os * per_sym) >> 4) + (uint32_t)(t % LPR));
                // label 4 s + b of block tb sits at stream position p = 16 tb + 4 s + b = d NT + a
                const int tb = t % LPR;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t wd = dw.w[s] & (mask * 0x01010101u);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int p = 16 * tb + 4 * s + b, d = p / NT, a = p % NT;       // (a, and d mod (4 / NT), are compile-time)
                        const int k = d ^ 128;
                        lab_mine[((rz * NT + a) * 16 + (k & 15)) * 16 + (k >> 4)] = (unsigned char)(wd >> (8 * b));
                    }
                }
                walk_wave_order();
                L = *reinterpret_cast<const uint4*>(lab_mine + t * 16);
                walk_wave_order();
            }
            cx<T> v[16];
            {
                const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const uint32_t lb = (wds[u >> 2] >> (8 * (u & 3))) & 0xFFu;
                    if constexpr (ABL & 32) v[u] = mk<T>((T)lb, (T)1);
                    else v[u] = s_txtab[lb];
                }
            }
            // ---- transmit transform: pass 1 (DIF spans 64, 16), transposition (a, g | u) -> (a, h | c), pass 2 (spans 4, 1) ----
            if constexpr (!(ABL & 64)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[g * m];
                    tw.a2[m - 1] = g_tw[4 * g * m];
                }
                r16_pass<T, true, false, 0, false, true, true>(nullptr, nullptr, 0, tw, nullptr, 0, v, v);
            }
            {
                const int ln = opaque(lane);
                const int a = ln >> 4, g = ln & 15;
                const int wbase = fw_slot(a, g), rbase = fw_slot(a, 16 * g);
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
                walk_wave_order();
            }
            if constexpr (!(ABL & 64)) {
                R16Tw64<T> none;
                r16_pass<T, true, false, 0, false, true, true, false, true>(nullptr, nullptr, 0, none, nullptr, 0, v, v);
            }
            // ---- noise words: lanes l and l ^ 4 hold the two samples of every pair; l evaluates the blocks of registers
            //      8 par .. 8 par + 7 (par = the parity of its sample times), keeps its half, hands over the other ----
            {
                const int ln = opaque(lane);
                const int r = (ln >> 4) % NR, h = ln & 15, par = (h >> 2) & 1;
                const uint64_t i0 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + (uint64_t)(4 * (h & 3) + ((h >> 2) & 2));   // even
                uint2* wm = s_words + (8 * par) * 64 + ln;
                uint2* wp = s_words + (8 * par) * 64 + (ln ^ 4);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    // register c = 8 par + cc: block b(c) = i0 / 2 + (c & 3) * 32 + (c >> 2) * 8, and c & 3 = cc & 3, c >> 2 = 2 par + (cc >> 2)
                    Words4 b;
                    const uint32_t bi = (uint32_t)(i0 >> 1) + (uint32_t)((cc & 3) * 32 + (cc >> 2) * 8) + (uint32_t)par * 16u;
                    if constexpr (ABL & 128) b.w[0] = b.w[1] = b.w[2] = b.w[3] = bi;
                    else b = rng.block(STREAM_NOISE, bi);
                    const uint2 even = make_uint2(b.w[0], b.w[1]), odd = make_uint2(b.w[2], b.w[3]);
                    wm[cc * 64] = par ? odd : even;
                    wp[cc * 64] = par ? even : odd;
                }
                walk_wave_order();
            }
            // ---- channel: R_r = sum_a H[r][a] T_a + noise on v_mfma_f64_4x4x4 (pipeline_mimo_qw.hip: the lane maps); the word pairs
            //      of eight registers at a time (sixteen more live registers instead of thirty-two) ----
            {
                const T hre = hA.x, him = hA.y, nhim = -hA.y;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    uint2 nw[8];
                    const int ln = opaque(lane);
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) nw[cc] = s_words[(8 * half + cc) * 64 + ln];
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) {
                        const int c = 8 * half + cc;
                        cx<T> z;
                        if constexpr (ABL & 128) z = mk<T>((T)nw[cc].x, sigma);
                        else z = cn_words(nw[cc].x, nw[cc].y, sigma, s_bm);
                        if constexpr (ABL & 256) {
                            v[c] = cadd(v[c], z);
                        } else if constexpr (sizeof(T) == 8) {
                            T yr = __builtin_amdgcn_mfma_f64_4x4x4f64(hre, v[c].x, z.x, 0, 0, 0);
                            T yi = __builtin_amdgcn_mfma_f64_4x4x4f64(him, v[c].x, z.y, 0, 0, 0);
                            yr = __builtin_amdgcn_mfma_f64_4x4x4f64(nhim, v[c].y, yr, 0, 0, 0);
                            yi = __builtin_amdgcn_mfma_f64_4x4x4f64(hre, v[c].y, yi, 0, 0, 0);
                            v[c] = mk<T>(yr, yi);
                        } else {
                            v[c] = fw_contract<NA>(Hc, v[c], z);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                walk_wave_order();
            }
            // ---- receive transform: pass 2' (DIT spans 1, 4), transposition back, pass 1' (spans 16, 64) ----
            if constexpr (!(ABL & 512)) {
                R16Tw64<T> none;
                r16_pass<T, false, true, 0, false, true, true, false, true>(nullptr, nullptr, 0, none, nullptr, 0, v, v);
            }
            {
                const int ln = opaque(lane);
                const int a = ln >> 4, g = ln & 15;
                const int wbase = fw_slot(a, g), rbase = fw_slot(a, 16 * g);
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
                walk_wave_order();
            }
            if constexpr (!(ABL & 512)) {
                const int g = opaque(lane) & 15;
                R16Tw64<T> tw;
#pragma unroll
                for (int m = 1; m <= 3; ++m) {
                    tw.a1[m - 1] = g_tw[g * m];
                    tw.a2[m - 1] = g_tw[4 * g * m];
                }
                r16_pass<T, false, true, 0, false, true, true>(nullptr, nullptr, 0, tw, nullptr, 0, v, v);
            }
            // ---- decode: est_a[k] = sum_r G[a][r] Y_r[k], the contraction over the four lanes of a column again; stream a's sixteen
            //      estimates land in lane (a, g), whose register L still holds their labels; decisions four at a time ----
            if constexpr (!(ABL & 1024)) {
                const T gre = gA.x, gim = gA.y, ngim = -gA.y;
                const uint32_t wds[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    cx<T> e[4];
                    int tx[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int u = 4 * i + jj;
                        if constexpr (sizeof(T) == 8) {
                            T er = __builtin_amdgcn_mfma_f64_4x4x4f64(gre, v[u].x, 0.0, 0, 0, 0);
                            T ei = __builtin_amdgcn_mfma_f64_4x4x4f64(gim, v[u].x, 0.0, 0, 0, 0);
                            er = __builtin_amdgcn_mfma_f64_4x4x4f64(ngim, v[u].y, er, 0, 0, 0);
                            ei = __builtin_amdgcn_mfma_f64_4x4x4f64(gre, v[u].y, ei, 0, 0, 0);
                            e[jj] = mk<T>(er, ei);
                        } else {
                            e[jj] = fw_contract<NA>(Gc, v[u], mk<T>(0, 0));
                        }
                        tx[jj] = (int)((wds[i] >> (8 * jj)) & 0xFFu);
                    }
                    walk_decide<T, DEC, 4>(mp, s_table, nullptr, e, tx, se, be);
                }
            } else {
                se += (unsigned)(v[0].x + v[15].y == 0.5);
            }
        }
        if constexpr (RZ == 1) {
            se = wave_sum_u32(se);
            be = wave_sum_u32(be);
            if (lane == 0) wg_account(totals[w], se, be, skipped, rl0, sym_out, bit_out);
        } else {                                       // two realizations: the halves' sums land in lanes 0 and 32
            const unsigned s0 = wave_sum_u32(rz ? 0u : se), b0 = wave_sum_u32(rz ? 0u : be);
            const unsigned s1 = wave_sum_u32(rz ? se : 0u), b1 = wave_sum_u32(rz ? be : 0u);
            const bool sk1 = __builtin_amdgcn_readlane((int)skipped, 32) != 0, sk0 = __builtin_amdgcn_readfirstlane((int)skipped) != 0;
            if (lane == 0) {
                wg_account(totals[w], s0, b0, sk0, rl0, sym_out, bit_out);
                if (rl0 + 1 < count) wg_account(totals[w], s1, b1, sk1, rl0 + 1, sym_out, bit_out);
            }
        }
    }
    wg_flush_waves<4>(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym, (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
}

template <typename T, int NA, int WPS, int ABL = 0>
static int launch_mimo_ofdm_fw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                               mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int N = 256, NT = NA, NR = NA, kRec = d64_rec<NT, NR>(), RZ = 4 / NA;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const int dec = walk_dec_kind(ctx, mp);
    mp.grid.G = 0;
    const size_t tab_len = ((size_t)mp.M + 1) & ~(size_t)1;
    const size_t lds = (size_t)4 * fw_plane_bytes<T>() + (2 * tab_len + 4 * RZ * (kRec + 1)) * sizeof(cx<T>) +
                       (sizeof(T) == 8 ? (size_t)((kBmLdsDoubles + 1) & ~1) * sizeof(double) : 8) + 4 * kFwLabBytes;
    MCLE_REQUIRE(lds + 512 <= (size_t)160 * 1024, "full-wave MIMO-OFDM kernel: %zu B of LDS do not fit", lds);
    auto kern = k_run_mimo_ofdm_fw<T, NA, WDEC_SLICER, WPS, ABL>;
    switch (dec) {
        case WDEC_QAM_CERT: kern = k_run_mimo_ofdm_fw<T, NA, WDEC_QAM_CERT, WPS, ABL>; break;
        case WDEC_QUAD_CERT: kern = k_run_mimo_ofdm_fw<T, NA, WDEC_QUAD_CERT, WPS, ABL>; break;
        case WDEC_AXIS4_CERT: kern = k_run_mimo_ofdm_fw<T, NA, WDEC_AXIS4_CERT, WPS, ABL>; break;
        default: break;
    }
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > WPS) per_cu = WPS;                            // 256 threads = one wavefront per SIMD and workgroup
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 20;
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kRec * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL((k_mimo_filters_planar<T, N, NT, NR>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed,
                           first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, (n + 4 * RZ - 1) / (4 * RZ), 8, 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, pp, mp, seed, first + off, n, (const cx<T>*)tw,
                           (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// 0 = launched; MCLE_E_UNSUPPORTED = outside the envelope (the caller stays on the planar kernel)
int run_mimo_ofdm_fw(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (!(cfg->fft_size == 256 && cfg->nt == cfg->nr && (cfg->nt == 4 || cfg->nt == 2) && cfg->num_used == 256 && (cfg->cp_size & 1) == 0))
        return MCLE_E_UNSUPPORTED;
    if (ctx->M > 256) return MCLE_E_UNSUPPORTED;
    {
        ModemParams<double> mp = pipe_modem<double>(ctx, cfg->demod_method);
        if (walk_dec_kind(ctx, mp) == WDEC_GENERIC) return MCLE_E_UNSUPPORTED;     // no certificate: the planar kernel's candidate grid
    }
    const bool two = ctx->opt[MCLE_OPT_F64_THREADS] == 262;
    if (dtype == MCLE_F32) {
        // complex64: registers bounded for TWO wavefronts per SIMD (nothing spilled): 1.46e8 realizations/s at 4 x 4 and 3.25e8 at 2 x 2 against
        // 1.30 / 2.69e8 at three (34 - 62 spilled registers) and 1.05 / 2.33e8 at four; the planar kernel: 1.29 / 1.67e8.  262: three.
        if (cfg->nt == 2)
            return two ? launch_mimo_ofdm_fw<float, 2, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
                       : launch_mimo_ofdm_fw<float, 2, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        return two ? launch_mimo_ofdm_fw<float, 4, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
                   : launch_mimo_ofdm_fw<float, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    }
#ifdef MCLE_EXPERIMENTS
    switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
#define MCLE_FW_ABL(V_) case V_: if (cfg->nt == 4) return launch_mimo_ofdm_fw<double, 4, 3, V_>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit); break;
        MCLE_FW_ABL(32) MCLE_FW_ABL(64) MCLE_FW_ABL(128) MCLE_FW_ABL(256) MCLE_FW_ABL(512) MCLE_FW_ABL(1024) MCLE_FW_ABL(2016)
#undef MCLE_FW_ABL
        default: break;
    }
#endif
    if (cfg->nt == 2)
        return two ? launch_mimo_ofdm_fw<double, 2, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
                   : launch_mimo_ofdm_fw<double, 2, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    return two ? launch_mimo_ofdm_fw<double, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
               : launch_mimo_ofdm_fw<double, 4, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
}

}  // namespace mcle
