// pipeline_mimo_mfma.hip -- config 4 (4x4 Blast + OFDM-1024) with the transforms and the two 4x4 contractions on
// the matrix cores (gfx950 v_mfma_f32_16x16x4_f32 / v_mfma_f32_4x4x1_16b_f32: exact f32, an fmaf chain per
// output, issued on the MFMA pipe next to the VALU work of the same and the co-resident waves).
//
// Same link, same draw ledger and same results contract as k_run_mimo_ofdm (pipelines.hip; reference
// apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:394-466); f32, N = 1024, 4x4 only.
//
// FFT-1024 = 16 x 16 x 4.  A 16-point DFT of 16 independent groups is one real matrix product
//     [Re; Im](out) = [[Wr, -Wi], [Wi, Wr]] [Re; Im](in)
// after one radix-2 split on the VALU (s = x[e] + x[e+8], d = x[e] - x[e+8]: even outputs = DFT-8 of s, odd
// outputs = DFT-8 of d with W16^e folded into the matrix), i.e. 2 x (16x16 real) x (16 x 16 groups) = 8 MFMAs per
// (antenna, 16 groups) instead of 16.  Per wavefront and pass: 4 antennas x 8 MFMAs (32 cycles each).
//   DIF (transmit, inverse via the re<->im swap identity IDFT(x) = swap(DFT(swap(x)))):
//     P1  DFT-16 over n1 (positions 64 n1 + n2)        x W1024^{k1 n2}     wave w owns columns n2 in [16w, 16w+16)
//     P2  DFT-16 over m1 (positions 64 k1 + 4 m1 + m2) x W64^{j1 m2}       wave w owns rows k1 in [4w, 4w+4)
//     P3  DFT-4  over m2 (positions 64 k1 + 4 j1 + m2)                      thread = one butterfly, same rows
//     -> position 64 k1 + 4 j1 + j2 holds time sample k1 + 16 j1 + 256 j2
//   DIT (receive) is the transpose: P3' (DFT-4, x W64), P2' (DFT-16, x W1024), P1' (DFT-16) -> natural bin order.
// P3, the channel (R = H T + noise, as 4x4x1 MFMAs with the noise as the initial accumulator) and P3' are ONE
// register-resident stage; the Blast decode (G Y, 4x4x1 MFMAs), the demodulator and the error count consume P1''s
// accumulators directly.  LDS round trips per OFDM symbol: 5 (+ the symbol scatter) instead of 12.
//
// LDS: per antenna a re plane and an im plane of 1024 floats (im plane 16 dwords further: a wave's two planes hit
// complementary bank halves), position p stored at p ^ (f(p >> 6) << 2), f(k) = (k & 7) ^ ((k & 1) << 3): every
// access of the passes above is bank-conflict free (tests/test_fft16_layout.py replays all of them; the b128
// stores of the middle stage are 2-way, below their own issue cost).
#include "fft.hpp"
#include "fft16.hpp"
#include "qam_pack.hpp"
#include "mimo.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"

namespace mcle {

struct MimoParams {
    int cp, num_used, n_ofdm_sym;
    int mmse;
    double noise_var;
};

// The channel draw and the f64 receive filter of every realization, one per lane, in a launch of their own (until round 2
// thread 0..63 of the main kernel did this for the workgroup's next 64 realizations: 128 f64 registers of work matrices in
// the middle of a kernel sized for 168 VGPRs -- 40 of its 72 spilled registers).  Record: H[16], G[16] x FFT scale, flag.
constexpr int kMimoRec = 2 * 4 * 4 + 1;
__global__ __launch_bounds__(64) void k_mimo_filters(MimoParams pp, uint64_t seed, uint64_t first, uint64_t count,
                                                     float2* __restrict__ recs) {
    constexpr int NA = 4;
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const double rx_scale = sqrt((double)(pp.num_used + pp.cp)) / (double)kF16N;
    const Rng rng(seed, first + rl);
    float2* rec = recs + rl * kMimoRec;
    double2 H[NA][NA], G[NA][NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const float2 h = cn_sample<float>(rng, STREAM_CHAN, (uint64_t)(r * NA + a), 1.f);
            rec[r * NA + a] = h;
            H[r][a] = mk<double>((double)h.x, (double)h.y);
        }
    const bool ok = blast_filter<NA, NA>(H, pp.mmse ? pp.noise_var : 0.0, G);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < NA; ++r)
            rec[NA * NA + a * NA + r] = make_float2((float)(G[a][r].x * rx_scale), (float)(G[a][r].y * rx_scale));
    rec[2 * NA * NA] = make_float2(ok ? 0.f : 1.f, 0.f);
}

// WAVES: waves per SIMD the register allocation is sized for (= workgroups per CU); FLAGS bit 0: draw the noise under
// the P1 / P2 MFMAs instead of inside the middle stage
template <int WAVES, int FLAGS>
__global__ __launch_bounds__(kPipeBlock, WAVES) void k_run_mimo_ofdm_mfma(MimoParams pp, ModemParams<float> mp,
                                                                       uint64_t seed, uint64_t first, uint64_t count,
                                                                       const float2* __restrict__ g_tw,
                                                                       const float2* __restrict__ g_recs, mcle_counters* counters,
                                                                       uint32_t* __restrict__ sym_out,
                                                                       uint32_t* __restrict__ bit_out) {
    constexpr int N = kF16N, NA = 4;
    constexpr int kRec = 2 * NA * NA + 1;                               // H, G, skip flag
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_d = reinterpret_cast<float*>(smem);                        // [NA][re plane | im plane]
    float2* s_txtab = reinterpret_cast<float2*>(s_d + NA * kF16Ant);    // [kMaxTable] constellation x tx scale
    float2* s_rec = s_txtab + kMaxTable;                                // [2][kRec + 1] H, G, flag of realization it & 1
    float4* s_tab4 = reinterpret_cast<float4*>(s_rec + 2 * (kRec + 1)); // [kMaxTable] {re, im, |c|^2/2, 0}
    unsigned* s_part = reinterpret_cast<unsigned*>(s_tab4 + kMaxTable); // [2][4 waves][2] error partials of realization it & 1
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_part + 16);
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [NA * num_used]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4, gb = g >> 1;
    const int U = pp.num_used, cp = pp.cp;
    const int per_sym = U * NA;
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const float sigma = (float)sqrt(pp.noise_var);
    const float tx_scale = (float)(1.0 / sqrt((double)NA) / sqrt((double)(U + cp)));
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const uint32_t mask4 = mask * 0x01010101u;
    const bool slicer = mp.method == MCLE_DEMOD_QAM_SLICER;   // s_idx then holds LEVEL bytes instead of labels
    const QamPack qp = qam_pack(mp);

    for (int m = tid; m < mp.M; m += kPipeBlock) {
        const float2 c = mp.g_table[m];
        s_txtab[m] = make_float2(c.x * tx_scale, c.y * tx_scale);
        s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
    }
    load_grid(mp, s_grid);
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    // ---- per-thread constants: DFT matrices, twiddles, LDS offsets ---------------------------------------------
    const Dft16Mats mats = dft16_mats(g_tw, lane);
    const int n2 = 16 * w + j;                       // P1 / P1': this lane's column
    const int k1p = 4 * w + (j >> 2), m2p = j & 3;   // P2 / P2': this lane's group (row k1p, residue m2p)
    // middle stage: lane -> butterfly (row k1m, j1m); lanes l and l ^ 1 hold time samples m and m + 1
    const int kkm = ((lane >> 5) << 1) | (lane & 1), j1m = (lane >> 1) & 15, k1m = 4 * w + kkm, par = lane & 1;
    // FLAGS bit 2: the three sets of pass twiddles are fetched from the (L1-resident) table at the top of their pass
    // instead of living in 24 registers across the whole loop (`opaque` keeps the loads from being hoisted)
    constexpr bool kReloadTw = (FLAGS & 4) != 0;
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
    if constexpr (!kReloadTw) {
        load_tw1a(tw1a, n2);
        load_tw2a(tw2a, m2p);
        load_tw1b(tw1b, k1p);
    }
#pragma unroll
    for (int m2 = 1; m2 < 4; ++m2) {
        tw2b[m2 - 1] = g_tw[(16 * m2 * j1m) & 1023];                          // P3': W64^{m2 j1} x slot order
        if (par && (m2 & 1)) tw2b[m2 - 1] = make_float2(-tw2b[m2 - 1].x, -tw2b[m2 - 1].y);
    }
    // dword offsets inside an antenna block (re plane; + kF16Plane for the im plane)
    const int plane_g = (g & 1) * kF16Plane;                                  // this lane's operand part: re / im plane
    const int p1_ld = 64 * gb + (n2 ^ (gb * 36));                             // element 2t+gb: (p1_ld ^ ((2t & 7) << 2)) + 128 t
    const int p1_st = 256 * g + (n2 ^ (16 * (g & 1)));                        // output 4g+x : (p1_st ^ ((x << 2) ^ ((x & 1) << 5))) + 64 x
    const int p2_base = 64 * k1p + (m2p | f16_swz(k1p));
    const int p2_ld = p2_base ^ (4 * gb);                                     // element 2t+gb: p2_ld ^ (8 t)
    const int p2_st = p2_base ^ (16 * g);                                     // output 4g+x : p2_st ^ (4 x)
    const int mid_off = 64 * k1m + ((4 * j1m) ^ f16_swz(k1m));                // 4 consecutive positions of the butterfly
    // symbol scatter, full band: lane (row e = lane >> 2, run = lane & 3) of wave w fills bins 64 e + 16 w + 4 run + (0..3)
    // of every antenna -- the wave's OWN 16 columns, so scatter -> P1 and P1' -> next scatter stay inside the wavefront
    const int sc_bin = 64 * (lane >> 2) + 16 * w + 4 * (lane & 3);
    const int sc_off = f16_pos(sc_bin);
    const int sc_blk = ((sc_bin + N / 2) & (N - 1)) >> 2;                     // Philox DATA block of those 4 subcarriers

    // noise samples pair up in Philox blocks by even / odd sample index; lanes l, l^1 share blocks when the
    // realization's sample indices keep the parity of the time index
    const bool pair_ok = ((row & 1) == 0) && (((N + cp) & 1) == 0) && ((cp & 1) == 0);
    const bool full_band = (U == N);
    constexpr bool kEarlyNoise = (FLAGS & 1) != 0;
    uint64_t it = 0, rl_prev = 0;
    __syncthreads();
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x, ++it) {
        const Rng rng(seed, first + rl);
        const int buf = (int)(it & 1);
        // this realization's record -> s_rec[buf] (read after the next workgroup barrier; its previous reader,
        // realization it - 2, is two barriers behind)
        if (tid < kRec) s_rec[buf * (kRec + 1) + tid] = g_recs[rl * kRec + tid];
        const float2* s_H = s_rec + buf * (kRec + 1);
        const float2* s_G = s_H + NA * NA;
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            // ---- symbols -> bins, stored re<->im swapped (inverse transform by the swap identity) ----
            const uint64_t n_first = (uint64_t)os * per_sym;
            if (full_band) {   // one Philox block per lane: 4 consecutive bins x 4 antennas, this wave's columns
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)(n_first >> 4) + sc_blk);
                const uint4 idx4 = make_uint4(dw.w[0] & mask4, dw.w[1] & mask4, dw.w[2] & mask4, dw.w[3] & mask4);
                *reinterpret_cast<uint4*>(s_idx + 16 * sc_blk) =
                    slicer ? make_uint4(labels_to_levels(idx4.x, qp), labels_to_levels(idx4.y, qp),
                                        labels_to_levels(idx4.z, qp), labels_to_levels(idx4.w, qp))
                           : idx4;
                const uint32_t wv[4] = {idx4.x, idx4.y, idx4.z, idx4.w};
                float2 sym[4][NA];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int a = 0; a < NA; ++a) sym[c][a] = s_txtab[(wv[c] >> (8 * a)) & 0xFFu];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const f4 vr = {sym[0][a].y, sym[1][a].y, sym[2][a].y, sym[3][a].y};
                    const f4 vi = {sym[0][a].x, sym[1][a].x, sym[2][a].x, sym[3][a].x};
                    *reinterpret_cast<f4*>(s_d + a * kF16Ant + sc_off) = vr;
                    *reinterpret_cast<f4*>(s_d + a * kF16Ant + kF16Plane + sc_off) = vi;
                }
                wave_lds_sync();
            } else {           // partial band: zero fill + scatter in block order across the workgroup
                __syncthreads();
                for (int p = tid; p < NA * kF16Ant; p += kPipeBlock) s_d[p] = 0.f;
                __syncthreads();
                const uint64_t n_last = n_first + per_sym;
                for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += kPipeBlock) {
                    const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const uint64_t n = (blk << 4) + jj;
                        if (n >= n_first && n < n_last) {
                            const int tx = (int)((dw.w[jj >> 2] >> ((jj & 3) * 8)) & mask);
                            const int nl = (int)(n - n_first);
                            const int a = nl & 3, d = nl >> 2;
                            s_idx[nl] = (unsigned char)(slicer ? labels_to_levels((uint32_t)tx, qp) : (uint32_t)tx);
                            const float2 c = s_txtab[tx];
                            const int off = a * kF16Ant + f16_pos(ofdm_bin(d, N, U));
                            s_d[off] = c.y;
                            s_d[off + kF16Plane] = c.x;
                        }
                    }
                }
                __syncthreads();
            }
            // noise of the middle stage: sample index of (r, c) = r*row + os*(N+cp) + cp + m, m = k1 + 16 j1 + 256 c.
            // Drawn here, one Philox block per antenna iteration of P1 / P2, where the VALU is otherwise idle under
            // the MFMAs; lanes l, l^1 share each block (even / odd sample) and trade the halves by DPP.
            f4 yre[4], yim[4];               // [slot] over r; slot s holds c = s ^ 2 par
            const uint64_t i_base = (uint64_t)os * (N + cp) + cp + (uint64_t)(k1m + 16 * j1m);
            auto noise_block = [&](int r, int cc) {
                const uint64_t i = (uint64_t)r * row + i_base + 256u * (2 * par + cc);
                const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i >> 1));
                const uint32_t k0 = par ? b.w[2] : b.w[0], k1 = par ? b.w[3] : b.w[1];
                const uint32_t g0 = par ? b.w[0] : b.w[2], g1 = par ? b.w[1] : b.w[3];
                const float2 keep = cn_from_words(k0, k1, sigma);
                const float2 give = cn_from_words(g0, g1, sigma);
                yre[cc][r] = keep.x;
                yim[cc][r] = keep.y;
                yre[2 + cc][r] = dpp_swap1(give.x);
                yim[2 + cc][r] = dpp_swap1(give.y);
            };
            // ---- P1: DFT-16 over n1, x W1024^{k1 n2} ----
            if constexpr (kReloadTw) load_tw1a(tw1a, opaque(n2));
            if constexpr ((FLAGS & 2) != 0) {
                dft16_pass4(s_d, plane_g, mats, tw1a,
                            [&](int t) { return (p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t; },
                            [&](int x) { return (p1_st ^ ((x << 2) ^ ((x & 1) << 5))) + 64 * x; },
                            [&]() {
                                if (kEarlyNoise && pair_ok) {
#pragma unroll
                                    for (int r = 0; r < NA; ++r) noise_block(r, 0);
                                }
                            });
            } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                float* pa = s_d + a * kF16Ant;
                const float* pl = pa + plane_g;
                float b[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = pl[(p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t];
                float2 o[4];
                dft16_mfma(mats, b, o);
                if (kEarlyNoise && pair_ok) noise_block(a, 0);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float2 v = cmul_pk(o[x], tw1a[x]);
                    const int off = (p1_st ^ ((x << 2) ^ ((x & 1) << 5))) + 64 * x;
                    pa[off] = v.x;
                    pa[off + kF16Plane] = v.y;
                }
            }
            }
            __syncthreads();
            if (tid == 0 && os == 0 && it > 0) {   // every wave is past the previous realization: account it
                const unsigned* q = s_part + (buf ^ 1) * 8;
                wg_account(totals, q[0] + q[2] + q[4] + q[6], q[1] + q[3] + q[5] + q[7],
                           s_rec[(buf ^ 1) * (kRec + 1) + 2 * NA * NA].x != 0.f, rl_prev, sym_out, bit_out);
            }
            // ---- P2: DFT-16 over m1, x W64^{j1 m2} ----
            if constexpr (kReloadTw) load_tw2a(tw2a, opaque(m2p));
            if constexpr ((FLAGS & 2) != 0) {
                dft16_pass4(s_d, plane_g, mats, tw2a, [&](int t) { return p2_ld ^ (8 * t); },
                            [&](int x) { return p2_st ^ (4 * x); },
                            [&]() {
                                if (kEarlyNoise && pair_ok) {
#pragma unroll
                                    for (int r = 0; r < NA; ++r) noise_block(r, 1);
                                }
                            });
            } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                float* pa = s_d + a * kF16Ant;
                const float* pl = pa + plane_g;
                float b[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = pl[p2_ld ^ (8 * t)];
                float2 o[4];
                dft16_mfma(mats, b, o);
                if (kEarlyNoise && pair_ok) noise_block(a, 1);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float2 v = cmul_pk(o[x], tw2a[x]);
                    pa[p2_st ^ (4 * x)] = v.x;
                    pa[(p2_st ^ (4 * x)) + kF16Plane] = v.y;
                }
            }
            }
            wave_lds_sync();
            // ---- middle stage: P3 (DFT-4) -> channel R = H T + noise -> P3' (DFT-4 x W64) ----
            {
                float xr[4][NA], xi[4][NA];     // [slot][antenna]: transmitted time samples (slot s = c ^ 2 par)
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const f4 R = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + mid_off);
                    const f4 I = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + kF16Plane + mid_off);
                    const float t0r = R[0] + R[2], t0i = I[0] + I[2], t1r = R[0] - R[2], t1i = I[0] - I[2];
                    const float t2r = R[1] + R[3], t2i = I[1] + I[3];
                    const float t3r = I[1] - I[3], t3i = R[3] - R[1];     // (z1 - z3) * (-i)
                    // planes hold swap(x): true sample = (im plane, re plane)
                    xi[0][a] = t0r + t2r; xr[0][a] = t0i + t2i;
                    xi[1][a] = t1r + t3r; xr[1][a] = t1i + t3i;
                    xi[2][a] = t0r - t2r; xr[2][a] = t0i - t2i;
                    xi[3][a] = t1r - t3r; xr[3][a] = t1i - t3i;
                }
                if (!kEarlyNoise && pair_ok) {
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        noise_block(r, 0);
                        noise_block(r, 1);
                    }
                }
                if (!pair_ok) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = s ^ (2 * par);
#pragma unroll
                        for (int r = 0; r < NA; ++r) {
                            const float2 z = cn_sample<float>(rng, STREAM_NOISE, (uint64_t)r * row + i_base + 256u * c, sigma);
                            yre[s][r] = z.x;
                            yim[s][r] = z.y;
                        }
                    }
                }
                {
                    float hr[NA], hi[NA], nhi[NA];   // A operands: lane (l & 3) = receive antenna r
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        const float2 h = s_H[(lane & 3) * NA + a];
                        hr[a] = h.x;
                        hi[a] = h.y;
                        nhi[a] = -h.y;
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            yre[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(hr[a], xr[s][a], yre[s], 0, 0, 0);
                            yim[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(hi[a], xr[s][a], yim[s], 0, 0, 0);
                            yre[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(nhi[a], xi[s][a], yre[s], 0, 0, 0);
                            yim[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(hr[a], xi[s][a], yim[s], 0, 0, 0);
                        }
                }
                // P3': DFT-4 over the slots (their order is folded into tw2b), x W64^{m2 j1}
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const float t0r = yre[0][r] + yre[2][r], t0i = yim[0][r] + yim[2][r];
                    const float t1r = yre[0][r] - yre[2][r], t1i = yim[0][r] - yim[2][r];
                    const float t2r = yre[1][r] + yre[3][r], t2i = yim[1][r] + yim[3][r];
                    const float t3r = yim[1][r] - yim[3][r], t3i = yre[3][r] - yre[1][r];
                    const float2 v1 = cmul_pk(make_float2(t1r + t3r, t1i + t3i), tw2b[0]);
                    const float2 v2 = cmul_pk(make_float2(t0r - t2r, t0i - t2i), tw2b[1]);
                    const float2 v3 = cmul_pk(make_float2(t1r - t3r, t1i - t3i), tw2b[2]);
                    const f4 vr = {t0r + t2r, v1.x, v2.x, v3.x};
                    const f4 vi = {t0i + t2i, v1.y, v2.y, v3.y};
                    *reinterpret_cast<f4*>(s_d + r * kF16Ant + mid_off) = vr;
                    *reinterpret_cast<f4*>(s_d + r * kF16Ant + kF16Plane + mid_off) = vi;
                }
            }
            wave_lds_sync();
            // ---- P2': DFT-16 over j1, x W1024^{(4 m1 + m2) k1} ----
            if constexpr (kReloadTw) load_tw1b(tw1b, opaque(k1p));
            if constexpr ((FLAGS & 2) != 0) {
                dft16_pass4(s_d, plane_g, mats, tw1b, [&](int t) { return p2_ld ^ (8 * t); },
                            [&](int x) { return p2_st ^ (4 * x); }, [&]() {});
            } else {
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                float* pa = s_d + a * kF16Ant;
                const float* pl = pa + plane_g;
                float b[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = pl[p2_ld ^ (8 * t)];
                float2 o[4];
                dft16_mfma(mats, b, o);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const float2 v = cmul_pk(o[x], tw1b[x]);
                    pa[p2_st ^ (4 * x)] = v.x;
                    pa[(p2_st ^ (4 * x)) + kF16Plane] = v.y;
                }
            }
            }
            __syncthreads();
            // ---- P1': DFT-16 over k1 -> bins 64 n1 + n2 (n1 = 4g + x), then Blast decode, demodulate, count ----
            {
                float yr[4][NA], yi[4][NA];      // [x][receive antenna]
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const float* pl = s_d + r * kF16Ant + plane_g;
                    float b[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) b[t] = pl[(p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t];
                    float2 o[4];
                    dft16_mfma(mats, b, o);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        yr[x][r] = o[x].x;
                        yi[x][r] = o[x].y;
                    }
                }
                float gr[NA], gi[NA], ngi[NA];   // A operands: lane (l & 3) = stream a; G carries the FFT scale
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const float2 gg = s_G[(lane & 3) * NA + r];
                    gr[r] = gg.x;
                    gi[r] = gg.y;
                    ngi[r] = -gg.y;
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    f4 er = {0.f, 0.f, 0.f, 0.f}, ei = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        er = __builtin_amdgcn_mfma_f32_4x4x1f32(gr[r], yr[x][r], er, 0, 0, 0);
                        ei = __builtin_amdgcn_mfma_f32_4x4x1f32(gi[r], yr[x][r], ei, 0, 0, 0);
                        er = __builtin_amdgcn_mfma_f32_4x4x1f32(ngi[r], yi[x][r], er, 0, 0, 0);
                        ei = __builtin_amdgcn_mfma_f32_4x4x1f32(gr[r], yi[x][r], ei, 0, 0, 0);
                    }
                    const int d = ofdm_data_index(64 * (4 * g + x) + n2, N, U);
                    if (d >= 0) {
                        const uint32_t sent = *reinterpret_cast<const uint32_t*>(s_idx + 4 * d);
                        if (slicer) {
                            qam_count4(qam_levels4(er, ei, qp) ^ sent, qp, se, be);
                        } else {
                            float2 est[NA];
                            int dec[NA];
#pragma unroll
                            for (int a = 0; a < NA; ++a) est[a] = make_float2(er[a], ei[a]);
                            if (mp.grid.G > 0) {
                                demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_grid4_multi<NA>(s_tab4, s_grid, mp.grid, mp.M, est, d_); });
                            } else {
                                demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_mindist_multi<NA>(s_tab4, mp.M, est, d_); });
                            }
#pragma unroll
                            for (int a = 0; a < NA; ++a) {
                                const unsigned xo = ((sent >> (8 * a)) & 0xFFu) ^ (unsigned)dec[a];
                                se += (xo != 0u);
                                be += __popc(xo);
                            }
                        }
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
            const int buf = (int)((it - 1) & 1);
            const unsigned* q = s_part + buf * 8;
            wg_account(totals, q[0] + q[2] + q[4] + q[6], q[1] + q[3] + q[5] + q[7],
                       s_rec[buf * (kRec + 1) + 2 * NA * NA].x != 0.f, rl_prev, sym_out, bit_out);
        }
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym,
                 (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
    }
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (caller uses k_run_mimo_ofdm)
int run_mimo_ofdm_mfma(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                       mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (cfg->fft_size != kF16N || cfg->nt != 4 || cfg->nr != 4) return MCLE_E_UNSUPPORTED;
    if (ctx->opt[MCLE_OPT_NO_MFMA]) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(kF16N, MCLE_F32, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    const ModemParams<float> mp = pipe_modem<float>(ctx, cfg->demod_method);
    const size_t lds = (size_t)4 * kF16Ant * sizeof(float) + (size_t)(kMaxTable + 2 * 34) * sizeof(float2) +
                       kMaxTable * sizeof(float4) + 16 * sizeof(unsigned) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) + (size_t)4 * cfg->num_used + 16;
    // variants kept for A/B runs (MCLE_OPT_MFMA_VARIANT = 10 * waves + flags): 36 = 3 waves per SIMD, noise drawn in the
    // middle stage, operand loads of the four antennas first, pass twiddles fetched per pass (default: 7 spilled
    // registers, 1.535-1.544 ms per 65 536 realizations); 32 = the same with the twiddles resident (28 spilled, 1.555);
    // 30 = antenna-by-antenna passes; 21 = 2 waves per SIMD (229 VGPRs, no spills) with the noise under P1 / P2 (1.62)
    const int variant = ctx->opt[MCLE_OPT_MFMA_VARIANT] ? (int)ctx->opt[MCLE_OPT_MFMA_VARIANT] : 36;
    const int waves = variant / 10 == 2 ? 2 : 3;
    auto kern = variant == 30 ? k_run_mimo_ofdm_mfma<3, 0> : variant == 21 ? k_run_mimo_ofdm_mfma<2, 1>
                : variant == 32 ? k_run_mimo_ofdm_mfma<3, 2> : k_run_mimo_ofdm_mfma<3, 6>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > waves) per_cu = waves;     // __launch_bounds__(256, WAVES)
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 18;          // realizations per filter + link pair: bounds the record buffer (69 MB)
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kMimoRec * sizeof(float2), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL(k_mimo_filters, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed, first + off, n,
                           (float2*)recs);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, n);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const float2*)tw, (const float2*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

}  // namespace mcle
