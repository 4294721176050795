// pipeline_mimo_planar.hip -- config 4 (Blast + OFDM over a flat MIMO channel) on PLANAR samples: written in round 3 for
// complex128, the reference's own precision (apps/mimo/simulate_mimo.py:68-142, mimo/mimo.py:609-660,
// modulators/ofdm.py:394-466: complex128 throughout; the file was pipeline_mimo_f64.hip), a template over the scalar type since
// the end of round 4: the same kernels on planes of floats are the complex64 family and -- at (1024, 4x4) -- faster than the
// matrix-core kernel of pipeline_mimo_mfma.hip.  The notes below are the complex128 design notes.
//
// Same link, same draw ledger (philox.hpp) and same results contract as k_run_mimo_ofdm<double, N, NA> (pipelines.hip),
// whose per-realization counts it reproduces; what changed is how the f64 datapath and the LDS are used (round-3
// profile of that kernel: one wavefront per SIMD because 86 KiB of LDS allowed one workgroup per CU, VALU busy 0.43, half of
// all LDS cycles bank conflicts, 6 500 VALU instructions per wavefront and realization of which 3 400 were libm log /
// sincos):
//   * PLANAR samples: per antenna a re plane and an im plane of N doubles.  A complex128 element as one 16-byte
//     access runs into the b128 lane groups (16 lanes over 64 banks), which the radix-4 swizzle of fft.hpp was not made
//     for; as two 8-byte accesses per element every plane is an array of 8-byte slots, and lds_swz64 (below) keeps every
//     load AND store of every stage bank-conflict free.
//   * no LDS twiddle copy (16 KiB in f64 at N = 1024): twiddles come from the L1-resident global table, fetched one stage
//     ahead where a stage multiplies first, or live in registers (256-thread form).  64 KiB of planes + tables = 77 KiB
//     at N = 1024, 4 x 4 -> TWO workgroups per CU.
//   * the channel draw and the f64 receive filter of every realization in a launch of their own (k_mimo_filters_planar), like
//     the f32 matrix-core path.
//   * Box-Muller by table + short polynomial (bm_f64.hpp); min-distance decisions of a square QAM through the margin
//     certificate, of anything else through the candidate grid (both decision-identical to the sweep; the complex128
//     certificate by construction up to 2^11 level spacings and by a <= 1e-14-per-symbol bound beyond, modem.hpp / mcle.h).
// Round 4: a FAMILY, not a benchmark point -- fft_size in {256, 512, 1024, 2048} (a trailing radix-2 stage for 512 / 2048,
// like fft.hpp), Nt <= Nr in {2, 4} square plus the Nr > Nt shapes listed in run_mimo_ofdm_planar; the reference's OFDM /
// Blast take any of them (modulators/ofdm.py:52-94, mimo/mimo.py:264-309, :609-660).
// No matrix cores here, on purpose: v_mfma_f64_16x16x4_f64 issues in 65 cycles (2048 flops: 31.5 flop/clk/SIMD, measured,
// scripts/experiments/f64_rates.hip) against 4.8 cycles for a v_fma_f64 (26.7 flop/clk), does NOT overlap with VALU work
// of the same SIMD, and a dense DFT-16 needs 1024 flops where two radix-4 stages need 224 f64 instructions per 16 points:
// the matrix-core transform would cost 1.9 x the datapath time of the butterflies.
#include <type_traits>
#include <utility>

#include "fft.hpp"
#include "fft_r16.hpp"
#include "mimo.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"
#include "mimo_planar_common.hpp"

namespace mcle {

// N, NT x NR: the geometry.  AH = antennas per thread in the transform stages, TB = (N / 4) (NR / AH) threads per
// workgroup, WPS = wavefronts per SIMD the register allocation is bounded for (what the LDS lets share a CU).  The
// benchmark geometry (1024, 4 x 4): AH = 2 -> 512 threads, 4 wavefronts per SIMD at two workgroups per CU, 128 VGPRs;
// AH = 4 -> 256 threads, 2 per SIMD, up to 256 VGPRs, the twelve twiddles of a thread in registers: LDS caps the workgroups
// per CU at two, so the first form buys latency hiding with threads instead.
// VAR (MCLE_OPT_F64_VARIANT): 1, 2, 3 = timing bounds ONLY, the results are wrong by construction -- what fusing the channel
// stage into its neighbouring transform stages could save at most, DESIGN.md 5.5: bit 0 = the stores of the last transmit
// stage and of the channel stage dropped, bit 1 = the two workgroup barriers around the channel stage dropped.
// 4 = the radix-16 transforms above (256 threads, one transform per wavefront): correct results, same contract.
template <typename T, int N, int NT, int NR, int AH, int WPS, int VAR = 0>
__global__ __launch_bounds__((N / 4) * (NR / AH), WPS) void k_run_mimo_ofdm_planar(MimoParams pp, ModemParams<T> mp, uint64_t seed,
                                                                     uint64_t first, uint64_t count,
                                                                     const cx<T>* __restrict__ g_tw,
                                                                     const cx<T>* __restrict__ g_recs,
                                                                     mcle_counters* counters,
                                                                     uint32_t* __restrict__ sym_out,
                                                                     uint32_t* __restrict__ bit_out) {
    using SH = F64Shape<N>;
    static_assert(NT >= 1 && NT <= NR && NR % AH == 0 && N >= 256, "geometry");
    constexpr int kRec = d64_rec<NT, NR>(), NB = SH::NB, N4 = SH::N4;
    constexpr int TB = NB * (NR / AH), NW = TB / 64;                        // threads, wavefronts per workgroup
    // complex64: the H x and G y multiply-adds as two packed FMAs each (pk_cfma) instead of four scalar ones -- level at the benchmark
    // geometry WITHOUT its 32 spilled registers (123 registers), +21 % at 512 4x4, +38 % at 2048 4x4, -8 % at 256 4x4, which keeps the
    // scalar form (profiles/r05/c4_f32_packed_mac_ab.log)
    constexpr bool PKMAC = sizeof(T) == 4 && !(N == 256 && NT == 4);
    constexpr bool R16 = (VAR & 4) != 0;                                    // radix-16 passes, one transform per wavefront
    constexpr bool FUSED = R16 && (VAR & 8) != 0;                           // ... with pass C, the channel and pass C' as one stage
    constexpr bool EXACT = R16 && (VAR & 16) != 0;                          // ... with every layer-1 twiddle from the table
    // section ablation for the instruction-level account of the headline kernel (-DMCLE_EXPERIMENTS builds only, option
    // f64_variant = 32 .. : WRONG results by construction): 32 = no symbol draws / table look-ups in the scatter, 64 = no transmit
    // transform, 128 = no noise draws, 256 = no H x products, 512 = no receive transform, 1024 = no decode
    constexpr int ABL = VAR & ~31;
    static_assert(!R16 || (N == 1024 && NR == 4 && AH == 4), "radix-16 variant: 1024, four receive antennas, 256 threads");
    // a thread's stage twiddles in registers for the whole kernel: 1024 points with every antenna in one thread (round 3) and, since
    // the last day of round 6, complex64 at 2048 points (30 registers the 77 - 96-register kernels have; fetched stage by stage they
    // were 240 gathers per realization: +7 % at 2 x 2, +2 % at 4 x 4).  NOT the other sizes: at 256 / 512 the stage-ahead fetches
    // already hide, and the registers cost 5 - 9 % (complex64 512 2 x 2: 1.24 -> 1.13e8; complex128: level) -- profiles/r06/c4_pmc.log
    constexpr bool TWR = !R16 && ((AH == NR && N == 1024) || (N == 2048 && sizeof(T) == 4));
    auto swz = [](int e) { return R16 ? lds_swz16f(e) : lds_swz64(e); };
    static_assert(TB % 64 == 0 && TB <= 1024 && TB >= kRec && NW <= 16, "workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* s_d = reinterpret_cast<T*>(smem);                          // [NR][re plane | im plane][N]
    cx<T>* s_table = reinterpret_cast<cx<T>*>(s_d + 2 * NR * N);        // [tab_len] constellation
    cx<T>* s_txtab = s_table + ((mp.M + 1) & ~1);                         // [tab_len] constellation x tx scale
    cx<T>* s_rec = s_txtab + ((mp.M + 1) & ~1);                           // [2][kRec + 1]
    unsigned* s_part = reinterpret_cast<unsigned*>(s_rec + 2 * (kRec + 1)); // [2][16 waves][2]
    constexpr int kBm = std::is_same<T, double>::value ? ((kBmLdsDoubles + 1) & ~1) : 0;
    double* s_bm = reinterpret_cast<double*>(s_part + 64);        // [kBm] Box-Muller tables (complex128)
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_bm + kBm);
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [NT * num_used]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int U = pp.num_used, cp = pp.cp;
    const int per_sym = U * NT;
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NT) / sqrt((double)(U + cp)));
    const uint32_t mask = (uint32_t)(mp.M - 1);
    for (int m = tid; m < mp.M; m += TB) {
        const cx<T> c = mp.g_table[m];
        s_table[m] = c;
        s_txtab[m] = cscale(c, tx_scale);
    }
    load_grid(mp, s_grid);
    if constexpr (kBm != 0) bm_tables_to_lds(s_bm, tid, TB);
    __shared__ WgTotals totals;
    if (tid == 0) wg_zero(totals);

    const int bbt = tid & (NB - 1);                               // this thread's butterfly position
    const int grp = tid / NB;                                     // ... of antennas AH grp .. AH grp + AH - 1
    T* s_mine = s_d + grp * (2 * AH * N);
    const bool tx_grp = NT == NR || grp * AH < NT;                // wave-uniform: does this group transmit?
    TwRegs64<T, N> twr;
    if constexpr (TWR) twr = load_tw64<T, N>(g_tw, bbt);
    [[maybe_unused]] R16Tw64<T> tw16;
    // fused form: fetched at the top of each transform instead (24 register pairs the fused stage does not have to carry).  The
    // complex64 unfused form keeps them: fetched per transform (or per pass) its spills fall from 29 to 13 registers, all of them
    // loop invariants parked outside the hot loops, and the loads land on the critical path -- 5.19 against 4.88 ms.
    constexpr bool TW16_RELOAD = FUSED;
    if constexpr (R16 && !TW16_RELOAD) tw16 = load_r16_tw<T>(g_tw, lane);
    [[maybe_unused]] T* s_wave_re = s_d + (2 * w) * N;      // variant 4: wavefront w owns antenna w's transform
    [[maybe_unused]] T* s_wave_im = s_wave_re + N;
    uint64_t it = 0, rl_prev = 0;
    // the record of a realization is fetched one iteration ahead (one register pair per lane of the first wavefront):
    // loaded where it is parked, the global-memory latency sat in front of every realization's first barrier
    cx<T> rec_next = mk<T>(0, 0);
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
        const cx<T>* s_H = s_rec + buf * (kRec + 1);            // [NR][NT]
        const cx<T>* s_G = s_H + NT * NR;                       // [NT][NR]
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            // ---- transmit: symbols -> bins (Blast.encode's F-order split + OFDM subcarrier map) ----
            if (it > 0 || os > 0) __syncthreads();            // the previous symbol's decode has read the planes
            if (U != N) {
                for (int p = tid; p < 2 * NT * N; p += TB) s_d[p] = 0.0;
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * per_sym;
            const uint64_t n_last = n_first + per_sym;
            // full band, symbol boundaries on DATA blocks: a block is the NT antennas of 16 / NT consecutive subcarriers
            // d0 .. (d0 a multiple of 16 / NT <= 8), whose bins differ from bin(d0) in bits 0-2 only, which the swizzle
            // leaves alone -- one bin and one swizzle per block instead of sixteen
            const bool aligned_scatter = (16 % NT == 0) && U == N && (per_sym & 15) == 0;
            for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += TB) {
                Words4 dw;
                if constexpr (ABL & 32) dw.w[0] = dw.w[1] = dw.w[2] = dw.w[3] = (uint32_t)blk;
                else dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (aligned_scatter) {
                    const int nl0 = (int)((blk << 4) - n_first);
                    const int pos0 = swz(ofdm_bin(nl0 / NT, N, U));
                    *reinterpret_cast<uint4*>(s_idx + nl0) = make_uint4(dw.w[0] & (mask * 0x01010101u), dw.w[1] & (mask * 0x01010101u),
                                                                        dw.w[2] & (mask * 0x01010101u), dw.w[3] & (mask * 0x01010101u));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const cx<T> c = (ABL & 32) ? mk<T>((T)tx, (T)1) : s_txtab[tx];
                        const int pos = pos0 ^ (j / NT);          // antenna j mod NT of subcarrier d0 + j / NT
                        s_d[(2 * (j % NT)) * N + pos] = c.x;
                        s_d[(2 * (j % NT) + 1) * N + pos] = c.y;
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
                        const int pos = swz(ofdm_bin(d, N, U));
                        s_d[(2 * a) * N + pos] = c.x;
                        s_d[(2 * a + 1) * N + pos] = c.y;
                    }
                }
            }
            __syncthreads();
            if (tid == 0 && os == 0 && it > 0) {   // every wave is past the previous realization: account it
                const unsigned* q = s_part + (buf ^ 1) * 32;
                unsigned ts = 0, tb = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    ts += q[2 * i];
                    tb += q[2 * i + 1];
                }
                wg_account(totals, ts, tb, s_rec[(buf ^ 1) * (kRec + 1) + 2 * NT * NR].x != 0.0, rl_prev, sym_out, bit_out);
            }
            // ---- IFFT: radix-4 DIF (+ the radix-2 stage), natural -> digit-reversed positions (a DIF stage multiplies LAST:
            //      its twiddle fetch hides behind its own butterflies -- fetching a stage ahead as the forward transform does
            //      measured no gain).  Antenna groups beyond Nt have nothing to send and only keep the barriers. ----
            if constexpr (R16) {
                if constexpr (TW16_RELOAD) tw16 = load_r16_tw<T>(g_tw, opaque(lane));
                if constexpr (!(ABL & 64))
                    if (NT == NR || w < NT) r16_dif<T, true, !FUSED, EXACT>(s_wave_re, s_wave_im, lane, tw16, g_tw);   // wavefront = antenna
                __syncthreads();
            } else
            static_for<N4>([&](auto stc) {
                constexpr int st = decltype(stc)::value, S = SH::span(st);
                if (tx_grp) r4_stage_planar<T, N, true, true, S, AH, TWR, (VAR & 1) && st + 1 == N4 && !SH::HAS2>(s_mine, twr, g_tw, opaque(bbt));
                if constexpr (st + 1 < N4 || SH::HAS2)
                    fft_stage_sync<TB>(S);             // wave-local once the 4 S points of a group sit in one wavefront
                else if constexpr (!(VAR & 2))
                    __syncthreads();
            });
            if constexpr (SH::HAS2) {
                if (tx_grp) r2_stage_planar<T, N, AH>(s_mine, opaque(bbt));
                __syncthreads();
            }
            // ---- channel: R = H T + noise on the samples that survive CP removal ----
            if constexpr (FUSED) {
                // ONE register stage: the span-1 butterflies of the transmit transform (pass C), the channel, and the span-1
                // butterflies of the receive transform (pass C').  A thread holds the four positions 4 g .. 4 g + 3 of all four
                // antennas; position 4 g + d carries time sample mb + 256 d.  Lanes l and l ^ 32 hold m and m + 1, i.e. the two
                // samples of every NOISE block: the lower half draws the blocks of receive antennas 0 and 1, the upper half those
                // of 2 and 3, and one v_permlane32_swap per word hands each half the samples it did not draw -- the draw ledger
                // is unchanged (every block computed once), two LDS passes and their address work are gone.
                const int ln = opaque(lane);
                const int h = (ln >> 5) & 1;
                const int g = (ln & 15) | (w << 4) | (h << 6) | (((ln >> 4) & 1) << 7);
                const int base_slot = swz(4 * g);                              // position 4 g + d sits at base_slot ^ d
                cx<T> x[NT][4];
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        x[a][d] = mk<T>(s_d[(2 * a) * N + (base_slot ^ d)], s_d[(2 * a + 1) * N + (base_slot ^ d)]);
#pragma unroll
                for (int a = 0; a < NT; ++a) r4_inplace<T, true>(x[a][0], x[a][1], x[a][2], x[a][3]);
                const int mb = fft_index_of_pos<N>(4 * g);
                const bool paired = (cp & 1) == 0;                             // every row's first kept sample on a block boundary
                cx<T> y[NR][4];
                auto mix = [&](int d, const cx<T> (&nz)[NR]) {               // y[.][d] = noise + H x[.][d]
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        cx<T> z = nz[r];
                        if constexpr (ABL & 256) {
                            z = cadd(z, x[r % NT][d]);
                        } else {
#pragma unroll
                            for (int a = 0; a < NT; ++a) z = CxOps<T>::template fma<PKMAC>(s_H[r * NT + a], x[a][d], z);
                        }
                        y[r][d] = z;
                    }
                };
                if (paired) {
                    static_for<4>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        __builtin_amdgcn_sched_barrier(0);                     // one position at a time: its four inputs die as its
                        cx<T> nz[NR];                                        // four outputs are born
#pragma unroll
                        for (int rr = 0; rr < NR / 2; ++rr) {
                            const uint64_t i0 = (uint64_t)(rr + (NR / 2) * h) * row + (uint64_t)os * (N + cp) + cp + mb + 256 * d;
                            cx<T> za, zb;
                            if constexpr (ABL & 128) {
                                za = zb = mk<T>((T)(uint32_t)i0, sigma);
                            } else {
                                const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                                za = cn_words(b.w[0], b.w[1], sigma, s_bm);             // the even sample: the lower half's
                                zb = cn_words(b.w[2], b.w[3], sigma, s_bm);             // the odd sample: the upper half's
                            }
                            swap32_pair(za.x, zb.x, nz[rr].x, nz[rr + NR / 2].x);
                            swap32_pair(za.y, zb.y, nz[rr].y, nz[rr + NR / 2].y);
                        }
                        mix(d, nz);
                    });
                } else {                                                       // odd prefix: unpaired draws, half of every block used
                    static_for<4>([&](auto dc) {
                        constexpr int d = decltype(dc)::value;
                        __builtin_amdgcn_sched_barrier(0);
                        cx<T> nz[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const uint64_t i0 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + mb + 256 * d;
                            const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                            const uint32_t x0 = (i0 & 1) ? b.w[2] : b.w[0], x1 = (i0 & 1) ? b.w[3] : b.w[1];
                            nz[r] = cn_words(x0, x1, sigma, s_bm);
                        }
                        mix(d, nz);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    r4_inplace<T, false>(y[r][0], y[r][1], y[r][2], y[r][3]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        s_d[(2 * r) * N + (base_slot ^ d)] = y[r][d].x;
                        s_d[(2 * r + 1) * N + (base_slot ^ d)] = y[r][d].y;
                    }
                }
            } else {
                constexpr int JT = (N / 2) / TB;                // channel iterations per thread: a compile-time count
                static_assert(JT * TB == N / 2, "channel loop");
#pragma unroll
                for (int jj = 0; jj < JT; ++jj) {
                    const int j = tid + jj * TB;
                    const int half = j / (N / 4), rest = j - half * (N / 4);
                    const int p0 = 2 * half * (N / 4) + rest, p1 = p0 + N / 4;
                    const int m0 = fft_index_of_pos<N>(p0);  // even; position p1 holds m0 + 1
                    const int q0 = swz(p0), q1 = swz(p1);
                    cx<T> x0[NT], x1[NT];
#pragma unroll
                    for (int a = 0; a < NT; ++a) {
                        x0[a] = mk<T>(s_d[(2 * a) * N + q0], s_d[(2 * a + 1) * N + q0]);
                        x1[a] = mk<T>(s_d[(2 * a) * N + q1], s_d[(2 * a + 1) * N + q1]);
                    }
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const uint64_t i0 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + m0;
                        cx<T> z0, z1;
                        if ((i0 & 1) == 0) {     // tables of the Box-Muller from this workgroup's LDS copy
                            const Words4 b = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                            z0 = cn_words(b.w[0], b.w[1], sigma, s_bm);
                            z1 = cn_words(b.w[2], b.w[3], sigma, s_bm);
                        } else {
                            const Words4 b0 = rng.block(STREAM_NOISE, (uint32_t)(i0 >> 1));
                            const Words4 b1 = rng.block(STREAM_NOISE, (uint32_t)((i0 + 1) >> 1));
                            z0 = cn_words(b0.w[2], b0.w[3], sigma, s_bm);
                            z1 = cn_words(b1.w[0], b1.w[1], sigma, s_bm);
                        }
#pragma unroll
                        for (int a = 0; a < NT; ++a) {
                            const cx<T> h = s_H[r * NT + a];       // wave-uniform address: an LDS broadcast
                            z0 = CxOps<T>::template fma<PKMAC>(h, x0[a], z0);
                            z1 = CxOps<T>::template fma<PKMAC>(h, x1[a], z1);
                        }
                        if constexpr (VAR & 1) {
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
            if constexpr (!(VAR & 2)) __syncthreads();
            // ---- FFT: (the radix-2 stage +) radix-4 DIT, digit-reversed -> natural bins ----
            if constexpr (R16) {
                if constexpr (TW16_RELOAD) tw16 = load_r16_tw<T>(g_tw, opaque(lane));
                if constexpr (!(ABL & 512)) r16_dit<T, false, !FUSED, EXACT>(s_wave_re, s_wave_im, lane, tw16, g_tw);
                __syncthreads();
            } else if constexpr (TWR) {
                if constexpr (SH::HAS2) {
                    r2_stage_planar<T, N, AH>(s_mine, opaque(bbt));
                    fft_stage_sync<TB>(2);
                }
                static_for<N4>([&](auto stc) {
                    constexpr int st = decltype(stc)::value, S = SH::span(N4 - 1 - st);
                    r4_stage_planar<T, N, false, false, S, AH, TWR>(s_mine, twr, g_tw, opaque(bbt));
                    if constexpr (st + 1 < N4)
                        fft_stage_sync<TB>(4 * S);     // the next stage's 4 S-thread groups read what was written here
                    else
                        __syncthreads();
                });
            } else {                                  // every stage's twiddles fetched while the previous stage runs
                cx<T> wpre[2][3];
                if constexpr (SH::HAS2) {
                    stage_tw_fetch<T, N, 2>(g_tw, opaque(bbt), wpre[0]);
                    r2_stage_planar<T, N, AH>(s_mine, opaque(bbt));
                    fft_stage_sync<TB>(2);
                }
                static_for<N4>([&](auto stc) {
                    constexpr int st = decltype(stc)::value, S = SH::span(N4 - 1 - st);
                    if constexpr (st + 1 < N4) stage_tw_fetch<T, N, 4 * S>(g_tw, opaque(bbt), wpre[(st + 1) & 1]);
                    r4_stage_planar<T, N, false, false, S, AH, TWR>(s_mine, twr, g_tw, opaque(bbt), S > 1 ? wpre[st & 1] : nullptr);
                    if constexpr (st + 1 < N4)
                        fft_stage_sync<TB>(4 * S);
                    else
                        __syncthreads();
                });
            }
            // ---- receive: Blast decode (G carries the FFT scale), demodulate, count ----
            if constexpr (!(ABL & 1024)) {
                for (int d = tid; d < U; d += TB) {
                    const int bin = swz(ofdm_bin(d, N, U));
                    cx<T> y[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) y[r] = mk<T>(s_d[(2 * r) * N + bin], s_d[(2 * r + 1) * N + bin]);
                    uint32_t sent = 0;
                    if constexpr (NT == 4) {
                        sent = *reinterpret_cast<const uint32_t*>(s_idx + 4 * d);
                    } else if constexpr (NT == 2) {
                        sent = *reinterpret_cast<const uint16_t*>(s_idx + 2 * d);
                    } else {
#pragma unroll
                        for (int a = 0; a < NT; ++a) sent |= (uint32_t)s_idx[NT * d + a] << (8 * a);
                    }
                    if constexpr (AH == NR && NT == NR) {   // 256-thread form (256 VGPRs): the streams searched in lockstep
                        cx<T> est[NT];
                        int dec[NT];
#pragma unroll
                        for (int a = 0; a < NT; ++a) {
                            est[a] = mk<T>(0, 0);
#pragma unroll
                            for (int r = 0; r < NR; ++r) est[a] = CxOps<T>::template fma<PKMAC>(s_G[a * NR + r], y[r], est[a]);
                        }
                        if (mp.method != MCLE_DEMOD_QAM_SLICER && mp.grid.G > 0) {
                            demod_multi_cert(mp, est, dec, [&](int (&d_)[NT]) {
                                if constexpr (std::is_same<T, double>::value) {
                                    demod_grid_multi<NT>(s_table, s_grid, mp.grid, mp.M, est, d_);
                                } else {
#pragma unroll
                                    for (int a = 0; a < NT; ++a) d_[a] = demod_grid(s_table, s_grid, mp.grid, mp.M, est[a]);
                                }
                            });
                        } else {
#pragma unroll
                            for (int a = 0; a < NT; ++a) dec[a] = demod_one<T>(mp, s_table, s_grid, est[a]);
                        }
#pragma unroll
                        for (int a = 0; a < NT; ++a) {
                            const unsigned x = ((sent >> (8 * a)) & 0xFFu) ^ (unsigned)dec[a];
                            se += (x != 0u);
                            be += __popc(x);
                        }
                    } else {                        // stream by stream (the lockstep form spilled 47 registers at the
#pragma unroll                                      // 128-register bound: 1.83e7 -> 1.60e7 realizations/s)
                        for (int a = 0; a < NT; ++a) {
                            cx<T> est = mk<T>(0, 0);
#pragma unroll
                            for (int r = 0; r < NR; ++r) est = CxOps<T>::template fma<PKMAC>(s_G[a * NR + r], y[r], est);
                            const int dec = demod_one<T>(mp, s_table, s_grid, est);
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
            s_part[buf * 32 + 2 * w] = se;
            s_part[buf * 32 + 2 * w + 1] = be;
        }
        rl_prev = rl;
    }
    __syncthreads();
    if (tid == 0) {
        if (it > 0) {
            const int buf = (int)((it - 1) & 1);
            const unsigned* q = s_part + buf * 32;
            unsigned ts = 0, tb = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                ts += q[2 * i];
                tb += q[2 * i + 1];
            }
            wg_account(totals, ts, tb, s_rec[buf * (kRec + 1) + 2 * NT * NR].x != 0.0, rl_prev, sym_out, bit_out);
        }
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym,
                 (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
    }
}

// one geometry: filters + link, sliced so that the record buffer stays bounded
template <typename T, int N, int NT, int NR, int AH, int WPS, int VAR = 0>
static int launch_mimo_ofdm_planar(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                                mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int kRec = d64_rec<NT, NR>(), TB = (N / 4) * (NR / AH);
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);     // certificate / candidate grid (pruned search)
    const size_t tab_len = ((size_t)mp.M + 1) & ~(size_t)1;
    const size_t lds = (size_t)2 * NR * N * sizeof(T) + (2 * tab_len + 2 * (kRec + 1)) * sizeof(cx<T>) +
                       64 * sizeof(unsigned) + (sizeof(T) == 8 ? (size_t)((kBmLdsDoubles + 1) & ~1) * sizeof(double) : 0) +
                       (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       (((size_t)NT * cfg->num_used + 15) & ~(size_t)15) + 16;
    MCLE_REQUIRE(lds + 512 <= (size_t)160 * 1024, "planar MIMO-OFDM kernel: %zu B of LDS do not fit (fft_size %d, %d receive antennas)",
                 lds, N, NR);
    auto kern = k_run_mimo_ofdm_planar<T, N, NT, NR, AH, WPS, VAR>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    const int by_waves = (WPS * 256) / TB > 0 ? (WPS * 256) / TB : 1;          // what __launch_bounds__ allocated registers for
    if (per_cu < 1) per_cu = 1;
    if (per_cu > by_waves) per_cu = by_waves;
    const uint64_t resident = (uint64_t)ctx->n_cu * per_cu;
    const uint64_t kSlice = 1ull << 18;     // realizations per filter + link pair: bounds the record buffer (138 MB)
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kRec * sizeof(cx<T>), &recs))) return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        hipLaunchKernelGGL((k_mimo_filters_planar<T, N, NT, NR>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, pp, seed,
                           first + off, n, (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        // (realizations per workgroup: >= 8, and enough for the counters' flush to disappear at the small shapes -- pipe_common.hpp)
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, resident, n, flush_min_units(8, sizeof(T) == 4 ? 25600 : 12800, (uint64_t)N * NR), 16);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TB), lds, ctx->stream, pp, mp, seed, first + off, n,
                           (const cx<T>*)tw, (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                           d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

// host side: 0 = launched, MCLE_E_UNSUPPORTED = outside this family's envelope (caller uses k_run_mimo_ofdm<double, ...>).
// Geometry table: (fft_size, Nt x Nr) -> antennas per thread AH, threads, workgroups per CU (LDS), wavefronts per SIMD:
//   1024  4x4   radix-16 form: 4 antennas / wave-per-antenna, 256 threads, 2 workgroups per CU, 2 wavefronts per SIMD
//  (1024  4x4   2   512   2   4  radix-4, option f64_threads=512)   2048  4x4   2  1024   1   4        512  4x4   2   256   3   3        256  4x4   2  128  5  3
//   1024  2x2   2   256   3   3        2048  2x2   2   512   2   4        512  2x2   2   128   5   3        256  2x2   2   64  8  2
// Nt < Nr (mimo/mimo.py:264-309 takes any): every 1 <= Nt <= Nr <= 4 at every size runs the Nr geometry (Nr = 3: three
// antennas per thread, one group); antenna groups past Nt idle in the IFFT.
// complex64 (round 4): the same kernels on planes of floats.  Half the LDS per workgroup, 80 - 140 registers per thread where
// complex128 takes 120 - 250: the radix-4 geometries keep the complex128 table's wavefronts-per-SIMD bound (doubling it spilled
// 10 - 77 registers in every 4-receive-antenna shape), the radix-16 form of the benchmark size runs four workgroups per CU
// instead of two.
// (last day of round 6: at 256 points, and at 512 with two receive antennas, that bound was the complex128 LDS limit carried over --
//  those complex64 kernels hold 78 - 93 registers and a quarter of the planes, and ran at 0.47 - 0.69 of their issue slots
//  (profiles/r06/c4_pmc.log): two more wavefronts per SIMD there, +5 ... +18 %.  NOT elsewhere: 512 points with three receive antennas
//  LOSE 14 - 37 % at four wavefronts per SIMD, 1024 / 2048 do not move -- profiles/r06/planar_f32_wps_ab.log)
#ifndef MCLE_PLANAR_F32_WPS_PLUS
#define MCLE_PLANAR_F32_WPS_PLUS 2
#endif
template <typename T> constexpr int planar_wps(int n, int nt, int nr, int w64) {
    if (sizeof(T) == 4 && w64 <= 3 && ((n == 256 && !(nt == 4 && nr == 4)) || (n == 512 && nr == 2))) return w64 + MCLE_PLANAR_F32_WPS_PLUS;
#ifndef MCLE_PLANAR_F64_256_WPS3
#define MCLE_PLANAR_F64_256_WPS3 0
#endif
    if (MCLE_PLANAR_F64_256_WPS3 && sizeof(T) == 8 && n == 256 && nr <= 3 && w64 == 2) return 3;   // (A/B on the last day: +1.5 ... -1.5 %, not adopted)
    return w64;
}

// pipeline_mimo_qw.hip: the quarter-wave kernel of the benchmark geometry (complex128; MCLE_E_UNSUPPORTED outside its envelope)
int run_mimo_ofdm_qw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);

// pipeline_mimo_fw.hip: one realization per wavefront at fft_size 256 (complex128, 4 x 4; MCLE_E_UNSUPPORTED outside its envelope)
int run_mimo_ofdm_fw(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);

// pipeline_mimo_pw.hip: 1 / NW of the time samples per wavefront at fft_size 512 (NW = 2) and 1024 (NW = 4), decode on the matrix cores
int run_mimo_ofdm_pw(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);

template <typename T>
static int run_mimo_ofdm_planar_t(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                                  mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr bool F64 = sizeof(T) == 8;
    const int n = cfg->fft_size, nt = cfg->nt, nr = cfg->nr;
#define MCLE_F64_GEOM(N_, NT_, NR_, AH_, WPS_)                                                                      \
    if (n == N_ && nt == NT_ && nr == NR_)                                                                          \
        return launch_mimo_ofdm_planar<T, N_, NT_, NR_, AH_, planar_wps<T>(N_, NT_, NR_, WPS_)>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    // (1024, 4 x 4), the benchmark geometry: radix-16 passes, one transform per wavefront, 256 threads (default since round 4:
    // 10.66 ms per 262 144 realizations against 11.65 for the 512-thread radix-4 form and 12.15 for the 256-thread one,
    // profiles/r04/c4_f64_r16_ab.log).  MCLE_OPT_F64_THREADS: 512 = radix-4, two antennas per thread; 256 = radix-4, four
    // antennas per thread, twiddles in registers.  MCLE_OPT_F64_VARIANT 1 .. 3: timing bounds on the 512-thread form.
    if (n == 256 && nt == nr && (nt == 4 || nt == 2)) {
        {   // (either arithmetic since the last day of round 6: complex64 contracts on the VALU, pipeline_mimo_fw.hip)
            // round 6: the FULL-WAVE kernel (pipeline_mimo_fw.hip: a realization is one wavefront -- the quarter-wave kernel's register
            // passes without its radix-4 exchange stage, channel AND decode on v_mfma_f64_4x4x4, no workgroup barrier).
            // MCLE_OPT_F64_THREADS = 261: the planar radix-4 form of rounds 3-5; 262: full-wave bounded for two wavefronts per SIMD.
            const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
            if (thr == 0 || thr == 260 || thr == 262) {
                const int rq = run_mimo_ofdm_fw(ctx, F64 ? MCLE_F64 : MCLE_F32, cfg, seed, first, count, d_counters, d_sym, d_bit);
                if (rq != MCLE_E_UNSUPPORTED) return rq;
            }
        }
    }
    if ((n == 512 || n == 2048) && nt == 4 && nr == 4) {
        if constexpr (F64) {
            // round 6: the HALF-WAVE / EIGHTH-WAVE kernel (pipeline_mimo_pw.hip, NW = 2 / 8).  MCLE_OPT_F64_THREADS = 261 (and, at 2048,
            // 512 / 1024): the planar radix-4 forms.
            const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
            if (thr == 0 || thr == 260 || thr == 262) {
                const int rq = run_mimo_ofdm_pw(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                if (rq != MCLE_E_UNSUPPORTED) return rq;
            }
        }
    }
    if (n == 1024 && nt == 4 && nr == 4) {
        if constexpr (F64) {
            {   // The default since the end of round 6: the quarter-wave decomposition with the DECODE on the matrix cores as well
                // (pipeline_mimo_pw.hip, NW = 4: 32.4 against 35.2 ms per 2^20 realizations for pipeline_mimo_qw.hip on the same box,
                // profiles/r06/pw_ab.log).  MCLE_OPT_F64_THREADS = 263: the same, explicit; 264: registers bounded for two wavefronts
                // per SIMD; 260 / 262: the quarter-wave kernel of pipeline_mimo_qw.hip (VALU decode, four bins per thread).
                const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
                if (thr == 0 || thr == 263 || thr == 264) {
                    const int rq = run_mimo_ofdm_pw(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                    if (rq != MCLE_E_UNSUPPORTED) return rq;
                }
            }
            // The default since round 6: the QUARTER-WAVE kernel (pipeline_mimo_qw.hip: samples in registers between the passes, three
            // workgroups per CU, the channel contraction on v_mfma_f64_4x4x4) -- 8.82 against 10.49 ms per 262 144 realizations
            // (profiles/r06/qw_ab.log).  Outside its envelope (partial band, odd prefix, a constellation without a certificate)
            // the launch stays on the planar forms below.  MCLE_OPT_F64_THREADS: 260 = quarter-wave (the default, explicit),
            // 262 = the same bounded for two wavefronts per SIMD, 261 = the planar radix-16 form that was the default until round 5.
            {
                const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
                if (thr == 0 || thr == 260 || thr == 262) {          // (0: outside the part-wave kernel's envelope nothing reaches this line)
                    const int rq = run_mimo_ofdm_qw(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                    if (rq != MCLE_E_UNSUPPORTED) return rq;
                }
            }
#ifdef MCLE_EXPERIMENTS     // the timing-bound variants give WRONG counters by construction: never in the product build
            switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
                case 1: return launch_mimo_ofdm_planar<T, 1024, 4, 4, 2, 4, 1>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                case 2: return launch_mimo_ofdm_planar<T, 1024, 4, 4, 2, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                case 3: return launch_mimo_ofdm_planar<T, 1024, 4, 4, 2, 4, 3>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                default: break;
            }
            // section ablation of the headline form (f64_variant = 32 | 64 | ... | 1024, any combination of single sections listed)
            switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
#define MCLE_ABL(V_) case V_: return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 2, 12 | V_>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
                MCLE_ABL(32) MCLE_ABL(64) MCLE_ABL(128) MCLE_ABL(256) MCLE_ABL(512) MCLE_ABL(1024) MCLE_ABL(2016) MCLE_ABL(1920) MCLE_ABL(384)
#undef MCLE_ABL
                default: break;
            }
#endif
            if (ctx->opt[MCLE_OPT_F64_THREADS] == 258)  // every layer-1 twiddle from the table (A/B: 10.95 ms against 10.62 -- the nine
                return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 2, 28>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);   // loads per pass cost more than the eight products)
        }
        if (ctx->opt[MCLE_OPT_F64_THREADS] == 256)
            return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        if (ctx->opt[MCLE_OPT_F64_THREADS] == 512)
            return launch_mimo_ofdm_planar<T, 1024, 4, 4, 2, 4>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        if constexpr (!F64) {
            // complex64: the SEPARATE channel stage at a four-wavefront register bound is the fast form (4.88 ms per 262 144
            // realizations; fused 5.08; at a three-wavefront bound 5.26 / 5.47; the matrix-core kernel 5.89 -- min-distance
            // demodulator, profiles/r04/c4_f32_planar_ab.log); 259 = fused, 257 = the three-wavefront bound (A/B)
            if (ctx->opt[MCLE_OPT_F64_THREADS] == 259)
                return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 4, 12>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            if (ctx->opt[MCLE_OPT_F64_THREADS] == 257)
                return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 3, 4>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 4, 4>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        } else {
            if (ctx->opt[MCLE_OPT_F64_THREADS] == 257)      // radix-16 passes with the unfused channel stage (A/B)
                return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 2, 4>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            return launch_mimo_ofdm_planar<T, 1024, 4, 4, 4, 2, 12>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        }
    }
    // every Nt <= Nr <= 4 at every size: Nr = 2 / 4 two antennas per thread, Nr = 3 three (one group)
#define MCLE_F64_SIZE(N_, W2_, W4_)                                                                                  \
    MCLE_F64_GEOM(N_, 1, 2, 2, W2_) MCLE_F64_GEOM(N_, 2, 2, 2, W2_)                                                  \
    MCLE_F64_GEOM(N_, 1, 3, 3, 2) MCLE_F64_GEOM(N_, 2, 3, 3, 2) MCLE_F64_GEOM(N_, 3, 3, 3, 2)                        \
    MCLE_F64_GEOM(N_, 1, 4, 2, W4_) MCLE_F64_GEOM(N_, 2, 4, 2, W4_) MCLE_F64_GEOM(N_, 3, 4, 2, W4_)
    MCLE_F64_SIZE(256, 2, 3) MCLE_F64_GEOM(256, 4, 4, 2, 3)
    MCLE_F64_SIZE(512, 3, 3) MCLE_F64_GEOM(512, 4, 4, 2, 3)
    if (n == 1024 && nr == 4 && ctx->opt[MCLE_OPT_F64_THREADS] == 0) {      // Nt < 4 at the benchmark size: the radix-16 form too
        constexpr int WR16 = F64 ? 2 : 4, VR16 = F64 ? 12 : 4;       // complex128: fused, 2 workgroups per CU; complex64: unfused, 4
        if (nt == 1) return launch_mimo_ofdm_planar<T, 1024, 1, 4, 4, WR16, VR16>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        if (nt == 2) return launch_mimo_ofdm_planar<T, 1024, 2, 4, 4, WR16, VR16>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        if (nt == 3) return launch_mimo_ofdm_planar<T, 1024, 3, 4, 4, WR16, VR16>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
    }
    MCLE_F64_SIZE(1024, 3, 4)
    // 2048 with four receive antennas: 1 024 threads (two antennas per thread) are four wavefronts per SIMD by themselves, i.e. a
    // 128-register bound that the complex128 form does not meet (36 / 22 spilled registers at 4 x 4 / 3 x 4, complex64 13 / 0).  Four
    // antennas per thread (512 threads, 256 registers, nothing spilled) is the default where it is the faster form -- 4 x 4 in both
    // arithmetics (+7 % / +4 %) and 3 x 4 in complex64 (+3 %); 3 x 4 in complex128 keeps the 1 024-thread form WITH its spills (7.38
    // against 7.70 ms per 65 536 realizations), and so does 2 x 4 in complex128 with 12 (5.75 against 6.74 ms; complex64 2 x 4 and 1 x 4: 2.97 against
    // 3.71 and 2.79 against 3.51 ms for the 512-thread form): profiles/r05/planar_2048_ab.log.  MCLE_OPT_F64_THREADS: 512 / 1024 force either.
    if (n == 2048 && nr == 4) {
        const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
        const bool four = thr == 512 || (thr != 1024 && (nt == 4 || !F64));       // complex64: every Nt; complex128: 4 x 4 only
        if (four) {
            if (nt == 4) return launch_mimo_ofdm_planar<T, 2048, 4, 4, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            if (nt == 3) return launch_mimo_ofdm_planar<T, 2048, 3, 4, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            if (nt == 2) return launch_mimo_ofdm_planar<T, 2048, 2, 4, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
            return launch_mimo_ofdm_planar<T, 2048, 1, 4, 4, 2>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
        }
    }
    MCLE_F64_SIZE(2048, 4, 4) MCLE_F64_GEOM(2048, 4, 4, 2, 4)
#undef MCLE_F64_SIZE
#undef MCLE_F64_GEOM
    return MCLE_E_UNSUPPORTED;
}

int run_mimo_ofdm_planar(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                         mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    if (ctx->opt[MCLE_OPT_F64_GENERIC]) return MCLE_E_UNSUPPORTED;
    return dtype == MCLE_F32 ? run_mimo_ofdm_planar_t<float>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit)
                             : run_mimo_ofdm_planar_t<double>(ctx, cfg, seed, first, count, d_counters, d_sym, d_bit);
}

}  // namespace mcle
