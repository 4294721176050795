// pipeline_mimo_f64.hip -- config 4 (4x4 Blast + OFDM-1024) in complex128, the reference's own precision
// (apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660, modulators/ofdm.py:394-466: complex128 throughout).
//
// Same link, same draw ledger (philox.hpp) and same results contract as k_run_mimo_ofdm<double, 1024, 4> (pipelines.hip),
// whose per-realization counts it reproduces; what changed is how the f64 datapath and the LDS are used (round-3
// profile of that kernel: one wavefront per SIMD because 86 KiB of LDS allowed one workgroup per CU, VALU busy 0.43, half of
// all LDS cycles bank conflicts, 6 500 VALU instructions per wavefront and realization of which 3 400 were libm log /
// sincos):
//   * PLANAR samples: per antenna a re plane and an im plane of 1024 doubles.  A complex128 element as one 16-byte
//     access runs into the b128 lane groups (16 lanes over 64 banks), which the radix-4 swizzle of fft.hpp was not made
//     for; as two 8-byte accesses per element every plane is an array of 8-byte slots, and lds_swz64 (below) keeps every
//     load AND store of every stage bank-conflict free.
//   * no LDS twiddle copy (16 KiB in f64): a thread runs the same butterfly position in every realization, so its twelve
//     twiddles are registers.  64 KiB of planes + tables = 73 KiB -> TWO workgroups per CU.
//   * the channel draw and the f64 receive filter of every realization in a launch of their own (k_mimo_filters_f64), like
//     the f32 matrix-core path.
//   * Box-Muller by table + short polynomial (bm_f64.hpp), the min-distance search through the candidate grid
//     (decision-identical to the sweep, modem.hpp).
// No matrix cores here, on purpose: v_mfma_f64_16x16x4_f64 issues in 65 cycles (2048 flops: 31.5 flop/clk/SIMD, measured,
// scripts/experiments/f64_rates.hip) against 4.8 cycles for a v_fma_f64 (26.7 flop/clk), does NOT overlap with VALU work
// of the same SIMD, and a dense DFT-16 needs 1024 flops where two radix-4 stages need 224 f64 instructions per 16 points:
// the matrix-core transform would cost 1.9 x the datapath time of the butterflies.
#include <type_traits>
#include "fft.hpp"
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

constexpr int kD64N = 1024, kD64NA = 4;
constexpr int kD64Rec = 2 * kD64NA * kD64NA + 1;     // H, G x FFT scale, skip flag

// LDS position of element e of a plane of doubles: the 8-byte-slot swizzle of fft.hpp (conflict free for the loads and the
// stores of every stage of this kernel: the four legs of the radix-4 butterflies at spans 256 .. 1, the channel's position
// pairs, scatter and decode; tests/test_f64_layout.py replays all of them).
__host__ __device__ __forceinline__ int lds_swz64(int e) { return lds_swz<true>(e); }

__global__ __launch_bounds__(64) void k_mimo_filters_f64(MimoParams pp, uint64_t seed, uint64_t first, uint64_t count,
                                                         double2* __restrict__ recs) {
    constexpr int NA = kD64NA;
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const double rx_scale = sqrt((double)(pp.num_used + pp.cp)) / (double)kD64N;
    const Rng rng(seed, first + rl);
    double2* rec = recs + rl * kD64Rec;
    double2 H[NA][NA], G[NA][NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            H[r][a] = cn_sample<double>(rng, STREAM_CHAN, (uint64_t)(r * NA + a), 1.0);
            rec[r * NA + a] = H[r][a];
        }
    const bool ok = blast_filter<NA, NA>(H, pp.mmse ? pp.noise_var : 0.0, G);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < NA; ++r) rec[NA * NA + a * NA + r] = mk<double>(G[a][r].x * rx_scale, G[a][r].y * rx_scale);
    rec[2 * NA * NA] = mk<double>(ok ? 0.0 : 1.0, 0.0);
}

// The three twiddles of this thread's butterfly position at the four spans that have any (256, 64, 16, 4; span 1 has
// none): w[j][q] = W^{(q+1) k N/(4s)}, k = tid mod s.  A DIF stage and the DIT stage of the same span use the same
// twelve values (conjugated for the inverse transform), so they live in 48 registers for the whole kernel -- fetched per
// stage from the global table they sat on the critical path of every stage (three dependent ~600-cycle loads at two
// wavefronts per SIMD).
struct TwRegs64 {
    double2 w[4][3];
};
__device__ __forceinline__ TwRegs64 load_tw64(const double2* __restrict__ g_tw, int tid) {
    TwRegs64 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = 256 >> (2 * j), k = tid & (s - 1), ts = kD64N / (4 * s);
#pragma unroll
        for (int q = 0; q < 3; ++q) t.w[j][q] = g_tw[(q + 1) * k * ts];
    }
    return t;
}

// one radix-4 butterfly position of every antenna, planar LDS.
// DIF (INV: the transmit IFFT): butterfly, then twiddle; DIT (forward FFT): twiddle, then butterfly.
// the three twiddles of butterfly position bb at span S, fetched from the (L1-resident) table
template <int S> __device__ __forceinline__ void stage_tw_fetch(const double2* __restrict__ g_tw, int bb, double2 (&w)[3]) {
    constexpr int ts = kD64N / (4 * S);
    const int k = bb & (S - 1);
    w[0] = g_tw[k * ts];
    w[1] = g_tw[2 * k * ts];
    w[2] = g_tw[3 * k * ts];
}
// pre: twiddles fetched ahead by the caller (512-thread form, forward transform: a DIT stage multiplies FIRST, so a fetch
// issued inside the stage sits on its critical path; issued one stage early it hides behind that stage's butterflies)
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
// `mid` runs between the stage's sixteen LDS loads and its butterflies: independent work (variant 1: a quarter of the
// realization's noise draws) for the wave's own LDS round trip to hide behind
template <bool DIF, bool INV, int S, int NA, bool TWR, bool MIDFIRST = false, typename Mid = NoMid, bool NOSTORE = false>
__device__ __forceinline__ void r4_stage_planar(double* s_d, const TwRegs64& tw, const double2* __restrict__ g_tw, int bb,
                                                const double2* pre = nullptr, Mid&& mid = Mid()) {
    constexpr int N = kD64N, s = S;
    const int k = bb & (s - 1), g = bb / s;
    const int e0 = g * 4 * s + k;
    int i0, i1, i2, i3;
    lds_swz_r4<true>(e0, s, i0, i1, i2, i3);           // one swizzle + three XORs with per-stage constants (fft.hpp)
    double2 w1 = mk<double>(1, 0), w2 = w1, w3 = w1;
    if (s > 1) {
        if constexpr (TWR) {                          // 256-thread form: the twelve twiddles are registers
            constexpr int j = S == 256 ? 0 : S == 64 ? 1 : S == 16 ? 2 : 3;
            w1 = tw.w[j][0];
            w2 = tw.w[j][1];
            w3 = tw.w[j][2];
        } else if (pre != nullptr) {
            w1 = pre[0];
            w2 = pre[1];
            w3 = pre[2];
        } else {                                      // 512-thread form (128 VGPRs): from the L1-resident table, per stage
            constexpr int ts = N / (4 * s);
            w1 = g_tw[k * ts];
            w2 = g_tw[2 * k * ts];
            w3 = g_tw[3 * k * ts];
        }
        if (INV) {
            w1.y = -w1.y;
            w2.y = -w2.y;
            w3.y = -w3.y;
        }
    }
    if constexpr (MIDFIRST) {                          // before the loads: the sixteen loaded values are not live beside it
        mid();
        __builtin_amdgcn_sched_barrier(0);
    }
    double xr[NA][4], xi[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const double* pr = s_d + (2 * a) * N;
        const double* pi = pr + N;
        xr[a][0] = pr[i0]; xr[a][1] = pr[i1]; xr[a][2] = pr[i2]; xr[a][3] = pr[i3];
        xi[a][0] = pi[i0]; xi[a][1] = pi[i1]; xi[a][2] = pi[i2]; xi[a][3] = pi[i3];
    }
    if constexpr (!MIDFIRST) mid();
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        double2 u0 = mk<double>(xr[a][0], xi[a][0]), u1 = mk<double>(xr[a][1], xi[a][1]),
                u2 = mk<double>(xr[a][2], xi[a][2]), u3 = mk<double>(xr[a][3], xi[a][3]);
        if (!DIF && s > 1) {
            u1 = cmul(u1, w1);
            u2 = cmul(u2, w2);
            u3 = cmul(u3, w3);
        }
        const double2 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = rot<double, INV>(csub(u1, u3));
        double2 y0 = cadd(a0, a2), y1 = cadd(a1, a3), y2 = csub(a0, a2), y3 = csub(a1, a3);
        if (DIF && s > 1) {
            y1 = cmul(y1, w1);
            y2 = cmul(y2, w2);
            y3 = cmul(y3, w3);
        }
        if constexpr (NOSTORE) {                      // timing bound only (variant 32): results computed, not stored
            asm volatile("" ::"v"(y0.x), "v"(y0.y), "v"(y1.x), "v"(y1.y), "v"(y2.x), "v"(y2.y), "v"(y3.x), "v"(y3.y));
            continue;
        }
        double* pr = s_d + (2 * a) * N;
        double* pi = pr + N;
        pr[i0] = y0.x; pr[i1] = y1.x; pr[i2] = y2.x; pr[i3] = y3.x;
        pi[i0] = y0.y; pi[i1] = y1.y; pi[i2] = y2.y; pi[i3] = y3.y;
    }
}

// AH = antennas per thread in the transform stages: 4 -> 256 threads per workgroup (2 wavefronts per SIMD at two
// workgroups per CU, up to 256 VGPRs), 2 -> 512 threads (4 wavefronts per SIMD, 128 VGPRs): LDS caps the workgroups per CU
// at two, so the second form buys latency hiding with threads instead.
// VAR (MCLE_OPT_F64_VARIANT, round-4 variants measured against the plain form): bit 0 = the noise of the
// realization (Philox + Box-Muller: a pure function of the index, 22 % of the time) is drawn inside the four twiddled
// stages of the transmit transform, one receive antenna per stage, between the stage's LDS loads and its butterflies,
// and parked in registers until the channel stage -- independent work inside each wave's own LDS round trip.
// bit 2 = the 256-thread form fetches its twiddles per stage like the 512-thread form (48 registers less);
// bit 4 = the Box-Muller's node angle and its cos / sin as one 32-byte LDS entry (one address for both reads);
// bit 5 = TIMING BOUND ONLY, wrong results: the stores of the last transmit stage and of the channel stage and the two
// barriers around the channel dropped -- an upper bound of what fusing the channel into its neighbours could save;
// bit 3 (with bit 0) = the draws BEFORE the stage's loads instead of behind them (the loaded values are not live beside them).
template <int AH, int VAR>
__global__ __launch_bounds__(256 * (4 / AH), 2 * (4 / AH)) void k_run_mimo_ofdm_f64(MimoParams pp, ModemParams<double> mp, uint64_t seed,
                                                                     uint64_t first, uint64_t count,
                                                                     const double2* __restrict__ g_tw,
                                                                     const double2* __restrict__ g_recs,
                                                                     mcle_counters* counters,
                                                                     uint32_t* __restrict__ sym_out,
                                                                     uint32_t* __restrict__ bit_out) {
    constexpr int N = kD64N, NA = kD64NA, kRec = kD64Rec;
    constexpr int TB = 256 * (NA / AH), NW = TB / 64;                       // threads, wavefronts per workgroup
    constexpr bool TWR = AH == NA && !(VAR & 4);                            // the twelve twiddles of a thread in registers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_d = reinterpret_cast<double*>(smem);                          // [NA][re plane | im plane][N]
    double2* s_table = reinterpret_cast<double2*>(s_d + 2 * NA * N);        // [tab_len] constellation
    double2* s_txtab = s_table + ((mp.M + 1) & ~1);                         // [tab_len] constellation x tx scale
    double2* s_rec = s_txtab + ((mp.M + 1) & ~1);                           // [2][kRec + 1]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_rec + 2 * (kRec + 1)); // [2][8 waves][2]
    double* s_bm = reinterpret_cast<double*>(s_part + 32);                  // [kBmLdsDoubles (+1)] Box-Muller tables
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_bm + ((kBmLdsDoubles + 1) & ~1));
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [NA * num_used]
    [[maybe_unused]] double* s_pk = reinterpret_cast<double*>(s_idx + ((4 * pp.num_used + 15) & ~15));   // variant 16: packed trig

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int U = pp.num_used, cp = pp.cp;
    const int per_sym = U * NA;
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const double sigma = sqrt(pp.noise_var);
    const double tx_scale = 1.0 / sqrt((double)NA) / sqrt((double)(U + cp));
    const uint32_t mask = (uint32_t)(mp.M - 1);
    for (int m = tid; m < mp.M; m += TB) {
        const double2 c = mp.g_table[m];
        s_table[m] = c;
        s_txtab[m] = cscale(c, tx_scale);
    }
    load_grid(mp, s_grid);
    bm_tables_to_lds(s_bm, tid, TB);
    if constexpr (VAR & 16) bm_trig_packed_to_lds(s_pk, tid, TB);
    // complex sample from two Philox words
    auto cn_words = [&](uint32_t x0, uint32_t x1, double sg) -> double2 {
        if constexpr (VAR & 16) {
            const double rad = sg * bm_sqrt(bm_neg_log(x0, s_bm));
            double sn, cs;
            bm_sincos_packed(x1, cs, sn, s_pk);
            return mk<double>(rad * cs, rad * sn);
        } else {
            return cn_from_words_lds(x0, x1, sg, s_bm);
        }
    };
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    const int bbt = tid & 255;                                   // this thread's butterfly position
    double* s_mine = s_d + (tid >> 8) * (2 * AH * N);             // ... of antennas AH (tid >> 8) ...
    TwRegs64 twr;
    if constexpr (TWR) twr = load_tw64(g_tw, bbt);
    uint64_t it = 0, rl_prev = 0;
    // the record of a realization is fetched one iteration ahead (one register pair per lane of the first wavefront):
    // loaded where it is parked, the global-memory latency sat in front of every realization's first barrier
    double2 rec_next = mk<double>(0, 0);
    if (tid < kRec && blockIdx.x < count) rec_next = g_recs[(uint64_t)blockIdx.x * kRec + tid];
    __syncthreads();
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x, ++it) {
        const Rng rng(seed, first + rl);
        const int buf = (int)(it & 1);
        // this realization's record -> s_rec[buf] (first read after the next workgroup barrier; its previous reader,
        // realization it - 2, is many barriers behind)
        if (tid < kRec) {
            s_rec[buf * (kRec + 1) + tid] = rec_next;
            if (rl + gridDim.x < count) rec_next = g_recs[(rl + gridDim.x) * kRec + tid];
        }
        const double2* s_H = s_rec + buf * (kRec + 1);
        const double2* s_G = s_H + NA * NA;
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            // ---- transmit: symbols -> bins (Blast.encode's F-order split + OFDM subcarrier map) ----
            if (it > 0 || os > 0) __syncthreads();            // the previous symbol's decode has read the planes
            if (U != N) {
                for (int p = tid; p < 2 * NA * N; p += TB) s_d[p] = 0.0;
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * per_sym;
            const uint64_t n_last = n_first + per_sym;
            // full band, symbol boundaries on DATA blocks: a block is the four antennas of four consecutive subcarriers
            // d0 .. d0 + 3 (d0 a multiple of 4), whose bins differ from bin(d0) in bits 0-1 only, which the swizzle leaves
            // alone -- one bin and one swizzle per block instead of sixteen
            const bool aligned_scatter = U == N && (per_sym & 15) == 0;
            for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += TB) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (aligned_scatter) {
                    const int nl0 = (int)((blk << 4) - n_first);
                    const int pos0 = lds_swz64(ofdm_bin(nl0 / NA, N, U));
                    *reinterpret_cast<uint4*>(s_idx + nl0) = make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u),
                                                                        dw.w[2] & (mask * 0x01010101u), dw.w[3] & (mask * 0x01010101u));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const double2 c = s_txtab[tx];
                        const int pos = pos0 ^ (j >> 2);          // NA = 4: antenna j & 3 of subcarrier d0 + (j >> 2)
                        s_d[(2 * (j & 3)) * N + pos] = c.x;
                        s_d[(2 * (j & 3) + 1) * N + pos] = c.y;
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
                        const double2 c = s_txtab[tx];
                        const int pos = lds_swz64(ofdm_bin(d, N, U));
                        s_d[(2 * a) * N + pos] = c.x;
                        s_d[(2 * a + 1) * N + pos] = c.y;
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && os == 0 && it > 0) {   // every wave is past the previous realization: account it
                const unsigned* q = s_part + (buf ^ 1) * 16;
                unsigned ts = 0, tb = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    ts += q[2 * i];
                    tb += q[2 * i + 1];
                }
                wg_account(totals, ts, tb, s_rec[(buf ^ 1) * (kRec + 1) + 2 * NA * NA].x != 0.0, rl_prev, sym_out, bit_out);
            }
            // ---- IFFT: radix-4 DIF, natural -> digit-reversed positions (a DIF stage multiplies LAST: its twiddle fetch hides
            //      behind its own butterflies -- fetching a stage ahead as the forward transform does measured no gain) ----
            // the two noise samples of receive antenna r at this thread's channel positions (iteration jj of the channel loop)
            auto noise_pair = [&](int jj, int r, double2& z0, double2& z1) {
                const int j = tid + jj * TB;
                const int half = j / (N / 4), rest = j - half * (N / 4);
                const int m0 = fft_index_of_pos<N>(2 * half * (N / 4) + rest);     // even; the partner position holds m0 + 1
                const uint64_t i0 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + m0;
                if ((i0 & 1) == 0) {     // tables of the Box-Muller from this workgroup's LDS copy
                    const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                    z0 = cn_words(b.w[0], b.w[1], sigma);
                    z1 = cn_words(b.w[2], b.w[3], sigma);
                } else {
                    const Words4 b0 = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                    const Words4 b1 = rng.block(STREAM_NOISE, (uint32_t)((i0 + 1) >> 1));
                    z0 = cn_words(b0.w[2], b0.w[3], sigma);
                    z1 = cn_words(b1.w[0], b1.w[1], sigma);
                }
            };
            constexpr int JT = (N / 2) / TB;                    // channel iterations per thread
            [[maybe_unused]] double2 nz[JT][NA][2];
            if constexpr (VAR & 1) {
                auto draw = [&](auto rc) {
                    return [&]() {
                        constexpr int r = decltype(rc)::value;
#pragma unroll
                        for (int jj = 0; jj < JT; ++jj) noise_pair(jj, r, nz[jj][r][0], nz[jj][r][1]);
                    };
                };
                r4_stage_planar<true, true, 256, AH, TWR, (VAR & 8) != 0>(s_mine, twr, g_tw, opaque(bbt), nullptr, draw(std::integral_constant<int, 0>()));
                __syncthreads();
                r4_stage_planar<true, true, 64, AH, TWR, (VAR & 8) != 0>(s_mine, twr, g_tw, opaque(bbt), nullptr, draw(std::integral_constant<int, 1>()));
                fft_stage_sync<TB>(64);
                r4_stage_planar<true, true, 16, AH, TWR, (VAR & 8) != 0>(s_mine, twr, g_tw, opaque(bbt), nullptr, draw(std::integral_constant<int, 2>()));
                fft_stage_sync<TB>(16);
                r4_stage_planar<true, true, 4, AH, TWR, (VAR & 8) != 0>(s_mine, twr, g_tw, opaque(bbt), nullptr, draw(std::integral_constant<int, 3>()));
                fft_stage_sync<TB>(4);
            } else {
                r4_stage_planar<true, true, 256, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                __syncthreads();
                r4_stage_planar<true, true, 64, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(64);
                r4_stage_planar<true, true, 16, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(16);
                r4_stage_planar<true, true, 4, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(4);
            }
            if constexpr (VAR & 32) {   // bound of variant "channel fused into the adjacent stages": WRONG RESULTS, timing only
                r4_stage_planar<true, true, 1, AH, TWR, false, NoMid, true>(s_mine, twr, g_tw, opaque(bbt));
            } else {
                r4_stage_planar<true, true, 1, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                __syncthreads();
            }
            // ---- channel: R = H T + noise on the samples that survive CP removal ----
            {
#pragma unroll
                for (int jj = 0; jj < JT; ++jj) {
                    const int j = tid + jj * TB;
                    const int half = j / (N / 4), rest = j - half * (N / 4);
                    const int p0 = 2 * half * (N / 4) + rest, p1 = p0 + N / 4;
                    const int q0 = lds_swz64(p0), q1 = lds_swz64(p1);
                    double2 x0[NA], x1[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        x0[a] = mk<double>(s_d[(2 * a) * N + q0], s_d[(2 * a + 1) * N + q0]);
                        x1[a] = mk<double>(s_d[(2 * a) * N + q1], s_d[(2 * a + 1) * N + q1]);
                    }
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        double2 z0, z1;
                        if constexpr (VAR & 1) {
                            z0 = nz[jj][r][0];
                            z1 = nz[jj][r][1];
                        } else {
                            noise_pair(jj, r, z0, z1);
                        }
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            const double2 h = s_H[r * NA + a];       // wave-uniform address: an LDS broadcast
                            z0 = cfma(h, x0[a], z0);
                            z1 = cfma(h, x1[a], z1);
                        }
                        if constexpr (VAR & 32) {
                            asm volatile("" ::"v"(z0.x), "v"(z0.y), "v"(z1.x), "v"(z1.y));
                        } else {
                            s_d[(2 * r) * N + q0] = z0.x;
                            s_d[(2 * r + 1) * N + q0] = z0.y;
                            s_d[(2 * r) * N + q1] = z1.x;
                            s_d[(2 * r + 1) * N + q1] = z1.y;
                        }
                    }
                }
            }
            if constexpr (!(VAR & 32)) __syncthreads();
            // ---- FFT: radix-4 DIT, digit-reversed -> natural bins ----
            if constexpr (TWR) {
                r4_stage_planar<false, false, 1, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(4);
                r4_stage_planar<false, false, 4, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(16);
                r4_stage_planar<false, false, 16, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                fft_stage_sync<TB>(64);
                r4_stage_planar<false, false, 64, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                __syncthreads();
                r4_stage_planar<false, false, 256, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
            } else {                                  // every stage's twiddles fetched while the previous stage runs
                double2 wa[3], wb[3];
                stage_tw_fetch<4>(g_tw, opaque(bbt), wa);
                r4_stage_planar<false, false, 1, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                stage_tw_fetch<16>(g_tw, opaque(bbt), wb);
                fft_stage_sync<TB>(4);
                r4_stage_planar<false, false, 4, AH, TWR>(s_mine, twr, g_tw, opaque(bbt), wa);
                stage_tw_fetch<64>(g_tw, opaque(bbt), wa);
                fft_stage_sync<TB>(16);
                r4_stage_planar<false, false, 16, AH, TWR>(s_mine, twr, g_tw, opaque(bbt), wb);
                stage_tw_fetch<256>(g_tw, opaque(bbt), wb);
                fft_stage_sync<TB>(64);
                r4_stage_planar<false, false, 64, AH, TWR>(s_mine, twr, g_tw, opaque(bbt), wa);
                __syncthreads();
                r4_stage_planar<false, false, 256, AH, TWR>(s_mine, twr, g_tw, opaque(bbt), wb);
            }
            __syncthreads();
            // ---- receive: Blast decode (G carries the FFT scale), demodulate, count ----
            {
                for (int d = tid; d < U; d += TB) {
                    const int bin = lds_swz64(ofdm_bin(d, N, U));
                    double2 y[NA];
#pragma unroll
                    for (int r = 0; r < NA; ++r) y[r] = mk<double>(s_d[(2 * r) * N + bin], s_d[(2 * r + 1) * N + bin]);
                    const uint32_t sent = *reinterpret_cast<const uint32_t*>(s_idx + 4 * d);
                    if constexpr (AH == NA) {       // 256-thread form (256 VGPRs): the four streams searched in lockstep
                        double2 est[NA];
                        int dec[NA];
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            est[a] = mk<double>(0, 0);
#pragma unroll
                            for (int r = 0; r < NA; ++r) est[a] = cfma(s_G[a * NA + r], y[r], est[a]);
                        }
                        if (mp.method != MCLE_DEMOD_QAM_SLICER && mp.grid.G > 0) {
                            demod_multi_cert(mp, est, dec, [&](int (&d_)[NA]) { demod_grid_multi<NA>(s_table, s_grid, mp.grid, mp.M, est, d_); });
                        } else {
#pragma unroll
                            for (int a = 0; a < NA; ++a) dec[a] = demod_one<double>(mp, s_table, s_grid, est[a]);
                        }
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            const unsigned x = ((sent >> (8 * a)) & 0xFFu) ^ (unsigned)dec[a];
                            se += (x != 0u);
                            be += __popc(x);
                        }
                    } else {                        // 512-thread form (128 VGPRs): stream by stream (the lockstep form
#pragma unroll                                      // spilled 47 registers there: 1.83e7 -> 1.60e7 realizations/s)
                        for (int a = 0; a < NA; ++a) {
                            double2 est = mk<double>(0, 0);
#pragma unroll
                            for (int r = 0; r < NA; ++r) est = cfma(s_G[a * NA + r], y[r], est);
                            const int dec = demod_one<double>(mp, s_table, s_grid, est);
                            const unsigned x = ((sent >> (8 * a)) & 0xFFu) ^ (unsigned)dec;
                            se += (x != 0u);
                            be += __popc(x);
                        }
                    }
                }
            }
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        if (lane == 0) {
            s_part[buf * 16 + 2 * w] = se;
            s_part[buf * 16 + 2 * w + 1] = be;
        }
        rl_prev = rl;
    }
    __syncthreads();
    if (tid == 0) {
        if (it > 0) {
            const int buf = (int)((it - 1) & 1);
            const unsigned* q = s_part + buf * 16;
            unsigned ts = 0, tb = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                ts += q[2 * i];
                tb += q[2 * i + 1];
            }
            wg_account(totals, ts, tb, s_rec[buf * (kRec + 1) + 2 * NA * NA].x != 0.0, rl_prev, sym_out, bit_out);
        }
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym,
                 (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
    }
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this kernel's envelope (caller uses k_run_mimo_ofdm<double, ...>)
int run_mimo_ofdm_f64(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                      mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (cfg->fft_size != kD64N || cfg->nt != 4 || cfg->nr != 4) return MCLE_E_UNSUPPORTED;
    if (ctx->opt[MCLE_OPT_F64_GENERIC]) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(kD64N, MCLE_F64, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    const ModemParams<double> mp = pipe_modem<double>(ctx, cfg->demod_method);     // with the candidate grid (pruned search)
    const size_t tab_len = ((size_t)mp.M + 1) & ~(size_t)1;
    const size_t lds = (size_t)2 * kD64NA * kD64N * sizeof(double) + (2 * tab_len + 2 * (kD64Rec + 1)) * sizeof(double2) +
                       32 * sizeof(unsigned) + (size_t)((kBmLdsDoubles + 1) & ~1) * sizeof(double) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       (((size_t)4 * cfg->num_used + 15) & ~(size_t)15) + 16 +
                       ((ctx->opt[MCLE_OPT_F64_VARIANT] & 16) ? (size_t)kBmPackedDoubles * sizeof(double) : 0);
    // MCLE_OPT_F64_THREADS: 0 / 512 = two antennas per thread, 512-thread workgroups (default); 256 = four antennas per thread
    const bool wide = ctx->opt[MCLE_OPT_F64_THREADS] != 256;
    const int var = (int)ctx->opt[MCLE_OPT_F64_VARIANT];
    auto kern = wide ? (var == 32 ? k_run_mimo_ofdm_f64<2, 32> : var == 16 ? k_run_mimo_ofdm_f64<2, 16> : var == 9 ? k_run_mimo_ofdm_f64<2, 9> : (var & 1) ? k_run_mimo_ofdm_f64<2, 1> : k_run_mimo_ofdm_f64<2, 0>)
                     : var == 13 ? k_run_mimo_ofdm_f64<4, 13> : var == 9 ? k_run_mimo_ofdm_f64<4, 9>
                     : var == 5 ? k_run_mimo_ofdm_f64<4, 5> : var == 4 ? k_run_mimo_ofdm_f64<4, 4>
                     : (var & 1) ? k_run_mimo_ofdm_f64<4, 1> : k_run_mimo_ofdm_f64<4, 0>;
    const int tb = wide ? 512 : 256;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;             // __launch_bounds__(256, 2)
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 18;     // realizations per filter + link pair: bounds the record buffer (138 MB)
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kD64Rec * sizeof(double2), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL(k_mimo_filters_f64, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed,
                           first + off, n, (double2*)recs);
        MCLE_LAUNCH_CHECK();
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, n);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(tb), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const double2*)tw, (const double2*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

}  // namespace mcle
