// pipelines.hip -- fused Monte Carlo pipelines: whole realizations generated, pushed through the
// link and scored on-chip; only integer counters leave the GPU.
//
//   run_awgn         C1  apps/awgn_modulators/simulate_psk.py:51-115
//   run_flat_fading  C2  SuChannel(JakesSampleGenerator) flat fading, y = h s + n, equalise y / h
//   run_ofdm_tdl     C3  notebooks/TDL_and_OFDM.ipynb OfdmTdlSimulator._run_simulation
//   run_mimo_ofdm    C4  apps/mimo/simulate_mimo.py:68-142 with per-antenna OFDM
//
// Randomness follows the mcle-philox-v1 contract (philox.hpp), keyed by the GLOBAL realization
// index, so counters do not depend on grid shape, batch split or GPU count.  Draw ledger:
//   DATA   symbol n of the realization (C4: n = c*Nt + a, the reference's flat index order)
//   CHAN   C4: H[r][a] = sample r*Nt + a
//   PHASE  C2/C3: phi[l][s] = uniform l*S + s, psi[l][s] = uniform L*S + l*S + s   (S taps)
//   NOISE  C1/C2: sample n; C3: sample j of the faded stream; C4: sample r*row + j, row = n_sym*(N+cp)
#include <cstdlib>

#include "fft.hpp"
#include "jakes.hpp"
#include "mimo.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"
#include "qam_pack.hpp"
#include "walk_f64.hpp"

namespace mcle {


// per-realization {sym, bit} error partials -> counter block (exact integer sums)
__global__ __launch_bounds__(256) void k_fold_counters(const unsigned* __restrict__ ws,
                                                       const unsigned* __restrict__ skipped, size_t n_real,
                                                       unsigned long long n_sym, unsigned long long n_bits,
                                                       mcle_counters* counters, uint32_t* __restrict__ sym_out,
                                                       uint32_t* __restrict__ bit_out) {
    unsigned long long se = 0, se2 = 0, be = 0, be2 = 0, ok = 0, sk = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_real; r += (size_t)gridDim.x * blockDim.x) {
        const bool skip = skipped && skipped[r];
        const unsigned long long s = ws[2 * r], t = ws[2 * r + 1];
        if (sym_out) sym_out[r] = skip ? 0xFFFFFFFFu : (uint32_t)s;
        if (bit_out) bit_out[r] = skip ? 0xFFFFFFFFu : (uint32_t)t;
        if (skip) {
            ++sk;
            continue;
        }
        ++ok;
        se += s;
        se2 += s * s;
        be += t;
        be2 += t * t;
    }
    if (!counters) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off, 64);
        se2 += __shfl_xor(se2, off, 64);
        be += __shfl_xor(be, off, 64);
        be2 += __shfl_xor(be2, off, 64);
        ok += __shfl_xor(ok, off, 64);
        sk += __shfl_xor(sk, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&counters->sym_errors, se);
        atomicAdd((unsigned long long*)&counters->sym_errors_sq, se2);
        atomicAdd((unsigned long long*)&counters->bit_errors, be);
        atomicAdd((unsigned long long*)&counters->bit_errors_sq, be2);
        atomicAdd((unsigned long long*)&counters->n_realizations, ok);
        atomicAdd((unsigned long long*)&counters->n_skipped, sk);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counters->n_symbols = n_sym;
        counters->n_bits = n_bits;
    }
}

// =================================================================================================
// C1 / C2: single-carrier links.  Work item = (realization, chunk of kChunk symbols); one workgroup
// per item, 16 symbols per thread per pass (one Philox DATA block + eight NOISE blocks).
// =================================================================================================
constexpr int kChunk = 16384;
constexpr int kMaxRays = 64;

struct FlatParams {
    int n_symbols;
    int L;             // 0: pure AWGN (h = 1); > 0: Jakes rays
    int rayleigh_iid;  // h ~ CN(0,1) per sample from the CHAN stream
    double Fd, t0, dt;
    double noise_sigma;
};

// LR > 0 (f32 only, fp.L == LR): the LR ray phasors are evaluated exactly (f64 phase) at the first of
// a lane's 16 consecutive symbols and advanced by one complex rotation per symbol after that
// (e^{j 2 pi w_l dt}, rounded once from f64); 15 rotations add < 1e-6 of phase error, below the
// v_sin/v_cos error.  LR == 0: every sample evaluated from the closed form (any L, and f64).
// (h s + z) / h, the flat-fading link followed by its one-tap equaliser (singleuser.py:130-151 and the notebooks' `/ h`).
// f64 (parity instantiation): literally that.  f32: s + z conj(h) / |h|^2 with one v_rcp_f32 -- the same value to
// rounding, a dozen instructions fewer per symbol (measured: 157 -> 145 VALU instructions per symbol row).
// (complex128, round 5: the same form as complex64 -- (h s + z) / h = s + z conj(h) / |h|^2, the product h s never formed -- with
//  the reciprocal as v_rcp_f64 + two Newton steps (rcp_newton, common.hpp; the IEEE division sequence the compiler emits for
//  1.0 / x is eleven instructions): 13 f64 instructions per symbol instead of 25, the value within a rounding of the reference's
//  complex division)
__device__ __forceinline__ double2 flat_equalised(double2 h, double2 s, double2 z) {
    const double d = fma(h.x, h.x, h.y * h.y);
    const double inv = rcp_newton(d);
    return mk<double>(fma(fma(z.x, h.x, z.y * h.y), inv, s.x), fma(fma(z.y, h.x, -(z.x * h.y)), inv, s.y));
}
__device__ __forceinline__ float2 flat_equalised(float2 h, float2 s, float2 z) {
    const float inv = __builtin_amdgcn_rcpf(fmaf(h.x, h.x, h.y * h.y));
    return make_float2(fmaf(fmaf(z.x, h.x, z.y * h.y), inv, s.x), fmaf(fmaf(z.y, h.x, -(z.x * h.y)), inv, s.y));
}

// MODE (f32): 0 one demod_one per symbol (any method), 1 packed level-domain slicer, 2 lockstep min-distance search
// MODE (f64, round 6): 10 + a decision form of walk_f64.hpp fixed at compile time (11 slicer, 12 QAM margin certificate, 13 quadrant,
// 14 on-axis certificate: walk_decide, four symbols at a time) -- the run-time demod_one per symbol with its method / certificate /
// grid switches kept 209 scalar registers spilled in this kernel; 0 = that generic form, for constellations without a certificate
template <typename T, int LR, int MODE>
__global__ __launch_bounds__(kPipeBlock, (sizeof(T) == 8 && LR == 8) ? 2 : 1) void k_run_flat(FlatParams fp, ModemParams<T> mp, uint64_t seed,
                                                         uint64_t first, uint64_t count, unsigned* __restrict__ ws) {
    constexpr bool kRec = LR > 0;   // ray phasors advanced by rotation inside a thread's run of 16 symbols (exact restart per run)
    __shared__ cx<T> s_table[kMaxTable];
    extern __shared__ unsigned long long s_grid[];       // [G*G] candidate grid (min-distance demodulation, f32)
    __shared__ double s_w[kMaxRays], s_psi[kMaxRays];
    __shared__ cx<T> s_rot[kMaxRays];
    __shared__ unsigned s_red[2 * (kPipeBlock / 64)];
    __shared__ float4 s_tab4[MODE == 2 ? kMaxTable : 1];     // {re, im, |c|^2 / 2, 0}: the lockstep searches of modem.hpp
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];   // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, kPipeBlock);
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    if constexpr (MODE == 2)
        for (int m = threadIdx.x; m < mp.M; m += kPipeBlock) {
            const float2 c = mp.g_table[m];
            s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
        }
    const int chunks = (fp.n_symbols + kChunk - 1) / kChunk;
    const uint64_t items = count * (uint64_t)chunks;
    const T sigma = (T)fp.noise_sigma;
    const T amp = fp.L > 0 ? (T)sqrt(1.0 / (double)fp.L) : (T)1;
    const uint32_t mask4 = (uint32_t)(mp.M - 1) * 0x01010101u;
    QamPack qp{};
    if constexpr (MODE == 1) qp = qam_pack(mp);
    // every workgroup takes a contiguous run of items: consecutive chunks of one realization share the ray set-up
    // (two f64 cosines / sincos per ray), which is redone only when the realization changes
    const uint64_t per_wg = (items + gridDim.x - 1) / gridDim.x;
    const uint64_t item_end = min((uint64_t)(blockIdx.x + 1) * per_wg, items);
    uint64_t last_rl = ~0ull;
    for (uint64_t item = (uint64_t)blockIdx.x * per_wg; item < item_end; ++item) {
        const uint64_t rl = item / chunks;
        const int chunk = (int)(item - rl * chunks);
        const Rng rng(seed, first + rl);
        __syncthreads();
        const bool fresh = rl != last_rl;
        last_rl = rl;
        if (fresh && fp.L > 0 && (int)threadIdx.x < fp.L) {
            // fading_generators.py:421-425: phi then psi, 2*pi*rand(L, 1, 1)
            const double two_pi = 6.283185307179586476925286766559;
            const double phi = two_pi * uniform_at(rng, STREAM_PHASE, threadIdx.x);
            const double psi = two_pi * uniform_at(rng, STREAM_PHASE, fp.L + threadIdx.x);
            if constexpr (sizeof(T) == 8) {
                const double w = two_pi * fp.Fd * cos(phi);
                s_w[threadIdx.x] = w;
                s_psi[threadIdx.x] = psi;
                double rs, rc;
                sincos(w * fp.dt, &rs, &rc);
                s_rot[threadIdx.x] = mk<T>(rc, rs);
            } else {
                const double w = fp.Fd * cos(phi);
                s_w[threadIdx.x] = w;
                s_psi[threadIdx.x] = psi / two_pi;
                double rs, rc;
                sincos(two_pi * (w * fp.dt), &rs, &rc);
                s_rot[threadIdx.x] = mk<T>((T)rc, (T)rs);
            }
        }
        __syncthreads();
        unsigned se = 0, be = 0;
        const int n_begin = chunk * kChunk;
        const int n_end = min(n_begin + kChunk, fp.n_symbols);
        // complex128 with the ray recurrence: a thread takes RUN = 4 consecutive DATA blocks (64 symbols) per pass and restarts its
        // phasors exactly (f64 sincos per ray) once per run instead of once per block: 8 sincos per 64 symbols instead of per 16
        // (64 rotations accumulate < 1e-14 of relative error); every other form keeps one block per pass
        // (a FULL chunk only: in the ragged last chunk of a realization 64-symbol runs would leave most of the workgroup idle)
        constexpr int RUNMAX = (kRec && sizeof(T) == 8) ? 4 : 1;
        const int RUN = (RUNMAX > 1 && n_end - n_begin == kChunk) ? RUNMAX : 1;
        for (int gr = n_begin + (int)threadIdx.x * 16 * RUN; gr < n_end; gr += kPipeBlock * 16 * RUN) {
          cx<T> ray[kRec ? LR : 1], rot[kRec ? LR : 1];
          for (int bq = 0; bq < RUN; ++bq) {
            const int g0 = gr + 16 * bq;
            if (g0 >= n_end) break;
            const Words4 dw = rng.block(STREAM_DATA, (uint32_t)(g0 >> 4));
            if (kRec && bq == 0) {
                const double t = jakes_time(fp.t0, fp.dt, (double)g0);
#pragma unroll
                for (int l = 0; l < (kRec ? LR : 1); ++l) {
                    ray[l] = jakes_ray<T>(s_w[l], s_psi[l], t);
                    if constexpr (sizeof(T) == 8) {
                        // complex128: the per-symbol rotations are wave-uniform -- as scalars (v_readfirstlane) they cost no vector
                        // registers; as 32 VGPRs next to the 32 of the ray phasors the kernel spilled 14 registers at its
                        // 256-register bound
                        const cx<T> v = s_rot[l];
                        rot[l] = mk<T>(__hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v.x)),
                                                        __builtin_amdgcn_readfirstlane(__double2loint(v.x))),
                                       __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v.y)),
                                                        __builtin_amdgcn_readfirstlane(__double2loint(v.y))));
                    } else {
                        rot[l] = s_rot[l];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {             // four symbols at a time: two whole noise blocks, one data word
                const int left = n_end - (g0 + 4 * q);
                if (left <= 0) break;
                cx<T> z[4], r[4];
                cn_pair_lds(rng, STREAM_NOISE, (uint32_t)((g0 >> 1) + 2 * q), sigma, z[0], z[1], s_bm);
                cn_pair_lds(rng, STREAM_NOISE, (uint32_t)((g0 >> 1) + 2 * q + 1), sigma, z[2], z[3], s_bm);
                const uint32_t dwt = dw.w[q] & mask4;
                cx<T> hq[4];                             // i.i.d. Rayleigh: the four channel samples are two whole CHAN blocks
                if (!kRec && fp.L == 0 && fp.rayleigh_iid) {
                    cn_pair<T>(rng, STREAM_CHAN, (uint32_t)((g0 >> 1) + 2 * q), (T)1, hq[0], hq[1]);
                    cn_pair<T>(rng, STREAM_CHAN, (uint32_t)((g0 >> 1) + 2 * q + 1), (T)1, hq[2], hq[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = g0 + 4 * q + e;
                    const cx<T> s = s_table[(dwt >> (8 * e)) & 0xFFu];
                    if (kRec) {
                        T hr = 0, hi = 0;
#pragma unroll
                        for (int l = 0; l < (kRec ? LR : 1); ++l) {
                            hr += ray[l].x;
                            hi += ray[l].y;
                            ray[l] = cmul(ray[l], rot[l]);
                        }
                        const cx<T> h = mk<T>(amp * hr, amp * hi);
                        r[e] = flat_equalised(h, s, z[e]);
                    } else if (fp.L > 0) {
                        const double t = jakes_time(fp.t0, fp.dt, (double)n);
                        T hr = 0, hi = 0;
                        for (int l = 0; l < fp.L; ++l) {
                            const cx<T> rr = jakes_ray<T>(s_w[l], s_psi[l], t);
                            hr += rr.x;
                            hi += rr.y;
                        }
                        const cx<T> h = mk<T>(amp * hr, amp * hi);
                        r[e] = flat_equalised(h, s, z[e]);
                    } else if (fp.rayleigh_iid) {
                        r[e] = flat_equalised(hq[e], s, z[e]);
                    } else {
                        r[e] = cadd(s, z[e]);
                    }
                }
                if constexpr (sizeof(T) == 8 && MODE >= 10) {
                    int tx[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tx[e] = (int)((dwt >> (8 * e)) & 0xFFu);
                        if (e >= left) r[e] = s_table[tx[e]];       // past the end of the realization: the point itself, no error
                    }
                    walk_decide<double, MODE - 10, 4>(mp, s_table, s_grid, r, tx, se, be);
                } else if constexpr (MODE == 1) {   // the four decisions in one packed level-domain slice (qam_pack.hpp)
                    const f4q re = {r[0].x, r[1].x, r[2].x, r[3].x}, im = {r[0].y, r[1].y, r[2].y, r[3].y};
                    uint32_t x = qam_levels4(re, im, qp) ^ labels_to_levels(dwt, qp);
                    if (left < 4) x &= (1u << (8 * left)) - 1u;
                    qam_count4(x, qp, se, be);
                } else if constexpr (MODE == 2) {   // the four symbols searched in lockstep (same decisions as demod_one)
                    int dec[4];
                    if (mp.M <= 8) demod_multi_cert(mp, r, dec, [&](int (&d_)[4]) { demod_mindist_multi<4>(s_tab4, mp.M, r, d_); });
                    else demod_multi_cert(mp, r, dec, [&](int (&d_)[4]) { demod_grid4_multi<4>(s_tab4, s_grid, mp.grid, mp.M, r, d_); });
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = e < left ? (unsigned)(((dwt >> (8 * e)) & 0xFFu) ^ (unsigned)dec[e]) : 0u;
                        se += (x != 0u);
                        be += __popc(x);
                    }
                } else {
                    int dec[4];
                    bool done = false;
                    if constexpr (sizeof(T) == 8) {   // complex128 grid search: the four symbols in lockstep
                        if (mp.method == MCLE_DEMOD_MINDIST && mp.grid.G > 0) {
                            demod_multi_cert(mp, r, dec, [&](int (&d_)[4]) { demod_grid_multi<4>(s_table, s_grid, mp.grid, mp.M, r, d_); });
                            done = true;
                        }
                    }
                    if constexpr (sizeof(T) == 8) {   // complex128 slicer: the clamp-before-floor form of the walks (walk_f64.hpp, round 6)
                        if (!done && mp.method == MCLE_DEMOD_QAM_SLICER) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) dec[e] = walk_qam_slicer(r[e], mp.qam_scale, mp.qam_L, mp.half_bits);
                            done = true;
                        }
                    }
                    if (!done) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) dec[e] = e < left ? demod_one(mp, s_table, s_grid, r[e]) : 0;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < left) {
                            const unsigned x = (unsigned)(((dwt >> (8 * e)) & 0xFFu) ^ (unsigned)dec[e]);
                            se += (x != 0u);
                            be += __popc(x);
                        }
                }
            }
          }
        }
        block_sum2(se, be, s_red);
        if (threadIdx.x == 0 && (se | be)) {
            atomicAdd(&ws[2 * rl], se);
            atomicAdd(&ws[2 * rl + 1], be);
        }
    }
}

// C2 on the matrix cores (f32, LR = 8 or 16 rays).  The 16 symbols of a group g share the ray phasors
// p_l = e^{j (2 pi w_l t_g + psi_l)} taken at the group's first symbol, and the channel at symbol i of the group is
//     h[i] = amp * sum_l e^{j 2 pi w_l dt i} p_l ,
// a real [16 x 2 LR] matrix -- constant over a realization, its entries rounded once from f64 -- times the
// [2 LR x groups] matrix of ray parts: v_mfma_f32_16x16x4_f32, 16 groups per product and LR / 2 products per plane, where
// k_run_flat<float, LR> spends one complex rotation and one complex add per ray and symbol (6 VALU instructions).
// Lane (j = lane & 15, b = lane >> 4) of a wavefront evaluates the ray parts 4 s + b (ray 2 s + (b >> 1); cosine or sine
// by b & 1) of the groups G0 + 16 t + j, t < 4, and receives h for the symbols 4 b .. 4 b + 3 of those four groups: 16
// symbols per lane and pass as in k_run_flat, in another order.  Data bytes: lane `lane` evaluates the Philox block of
// group G0 + lane and the 4 x 4 (word, row) transpose by v_permlane32_swap / v_permlane16_swap leaves word b of the four
// groups with lane (j, b); the noise of symbols 4 b .. 4 b + 3 is two whole Philox blocks.  Same draws and the same
// decisions as k_run_flat up to f32 rounding (reference: fading_generators.py:421-470, singleuser.py:130-151).
typedef float f4m __attribute__((ext_vector_type(4)));

template <int LR, int MODE>
__global__ __launch_bounds__(kPipeBlock) void k_run_flat_mfma(FlatParams fp, ModemParams<float> mp, uint64_t seed,
                                                              uint64_t first, uint64_t count, unsigned* __restrict__ ws) {
    constexpr int KS = LR / 2;                          // k-steps of four ray parts
    __shared__ float2 s_table[kMaxTable];
    __shared__ float4 s_tab4[MODE == 2 ? kMaxTable : 1]; // {re, im, |c|^2 / 2, 0}: the lockstep searches of modem.hpp
    extern __shared__ unsigned long long s_grid[];       // [G*G] candidate grid (min-distance demodulation)
    __shared__ double s_w[kMaxRays], s_psi[kMaxRays];
    __shared__ unsigned s_red[2 * (kPipeBlock / 64)];
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    if constexpr (MODE == 2)
        for (int m = threadIdx.x; m < mp.M; m += kPipeBlock) {
            const float2 c = mp.g_table[m];
            s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, b = lane >> 4;
    QamPack qp{};
    if constexpr (MODE == 1) qp = qam_pack(mp);
    const uint32_t mask4 = (uint32_t)(mp.M - 1) * 0x01010101u;
    const float sigma = (float)fp.noise_sigma;
    const double amp = sqrt(1.0 / (double)LR);
    const double two_pi = 6.283185307179586476925286766559;
    const int chunks = (fp.n_symbols + kChunk - 1) / kChunk;
    const uint64_t items = count * (uint64_t)chunks;
    const uint64_t per_wg = (items + gridDim.x - 1) / gridDim.x;
    const uint64_t item_end = min((uint64_t)(blockIdx.x + 1) * per_wg, items);
    uint64_t last_rl = ~0ull;
    double w[KS], psq[KS];        // this lane's rays: Doppler (Hz) and phase (turns; + 1/4 where the part is the cosine)
    float a_re[KS], a_im[KS];     // A operands: row j (symbol in the group) x part 4 s + b
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        w[s] = psq[s] = 0.0;
        a_re[s] = a_im[s] = 0.f;
    }
    for (uint64_t item = (uint64_t)blockIdx.x * per_wg; item < item_end; ++item) {
        const uint64_t rl = item / chunks;
        const int chunk = (int)(item - rl * chunks);
        const Rng rng(seed, first + rl);
        __syncthreads();
        const bool fresh = rl != last_rl;
        last_rl = rl;
        if (fresh && (int)threadIdx.x < LR) {
            // fading_generators.py:421-425: phi then psi, 2*pi*rand(L, 1, 1)
            const double phi = two_pi * uniform_at(rng, STREAM_PHASE, threadIdx.x);
            const double psi = two_pi * uniform_at(rng, STREAM_PHASE, LR + threadIdx.x);
            s_w[threadIdx.x] = fp.Fd * cos(phi);
            s_psi[threadIdx.x] = psi / two_pi;
        }
        __syncthreads();
        if (fresh) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int l = 2 * s + (b >> 1);
                w[s] = s_w[l];
                psq[s] = s_psi[l] + ((b & 1) ? 0.0 : 0.25);
                double sn, cs;
                sincos(two_pi * ((w[s] * fp.dt) * (double)j), &sn, &cs);
                a_re[s] = (float)(amp * ((b & 1) ? -sn : cs));
                a_im[s] = (float)(amp * ((b & 1) ? cs : sn));
            }
        }
        unsigned se = 0, be = 0;
        const int n_begin = chunk * kChunk;
        const int n_end = min(n_begin + kChunk, fp.n_symbols);
        for (int gw = n_begin + wave * 1024; gw < n_end; gw += kPipeBlock * 16) {    // symbols gw .. gw + 1023
            const Words4 dw = rng.block(STREAM_DATA, (uint32_t)((gw >> 4) + lane));
            uint32_t d[4] = {dw.w[0], dw.w[1], dw.w[2], dw.w[3]};
            {   // d[t] <- word b of the block evaluated by lane 16 t + j
                auto p = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                auto q = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                auto r0 = __builtin_amdgcn_permlane16_swap(p[0], q[0], false, false);
                auto r1 = __builtin_amdgcn_permlane16_swap(p[1], q[1], false, false);
                d[0] = r0[0];
                d[1] = r0[1];
                d[2] = r1[0];
                d[3] = r1[1];
            }
            // one tile of 16 groups per turn (rolled: the two accumulator chains of a tile already keep the matrix
            // pipe's dependent issues 64 cycles apart, and the body is 4 x smaller in the instruction cache)
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                const double tt = jakes_time(fp.t0, fp.dt, (double)(gw + 16 * (16 * t + j)));
                float bv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double x = fma(w[s], tt, psq[s]);
                    bv[s] = __builtin_amdgcn_sinf((float)__builtin_amdgcn_fract(x));
                }
                f4m hre = {0.f, 0.f, 0.f, 0.f}, him = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    hre = __builtin_amdgcn_mfma_f32_16x16x4f32(a_re[s], bv[s], hre, 0, 0, 0);
                    him = __builtin_amdgcn_mfma_f32_16x16x4f32(a_im[s], bv[s], him, 0, 0, 0);
                }
                const uint32_t dcur = d[0];
                d[0] = d[1];
                d[1] = d[2];
                d[2] = d[3];
                const int n0 = gw + 16 * (16 * t + j) + 4 * b;
                if (n0 >= n_end) continue;
                float2 z[4];
                cn_pair<float>(rng, STREAM_NOISE, (uint32_t)(n0 >> 1), sigma, z[0], z[1]);
                cn_pair<float>(rng, STREAM_NOISE, (uint32_t)(n0 >> 1) + 1u, sigma, z[2], z[3]);
                const uint32_t dwt = dcur & mask4;
                float2 r[4];
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    r[v] = flat_equalised(make_float2(hre[v], him[v]), s_table[(dwt >> (8 * v)) & 0xFFu], z[v]);
                const int left = n_end - n0;          // >= 1 symbols of this quad exist
                if constexpr (MODE == 1) {
                    const f4q re = {r[0].x, r[1].x, r[2].x, r[3].x}, im = {r[0].y, r[1].y, r[2].y, r[3].y};
                    uint32_t x = qam_levels4(re, im, qp) ^ labels_to_levels(dwt, qp);
                    if (left < 4) x &= (1u << (8 * left)) - 1u;
                    qam_count4(x, qp, se, be);
                } else {
                    int dec[4];
                    if constexpr (MODE == 2) {
                        if (mp.M <= 8) demod_multi_cert(mp, r, dec, [&](int (&d_)[4]) { demod_mindist_multi<4>(s_tab4, mp.M, r, d_); });
                        else demod_multi_cert(mp, r, dec, [&](int (&d_)[4]) { demod_grid4_multi<4>(s_tab4, s_grid, mp.grid, mp.M, r, d_); });
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v) dec[v] = demod_one(mp, s_table, s_grid, r[v]);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const unsigned x = v < left ? (unsigned)(((dwt >> (8 * v)) & 0xFFu) ^ (unsigned)dec[v]) : 0u;
                        se += (x != 0u);
                        be += __popc(x);
                    }
                }
            }
        }
        block_sum2(se, be, s_red);
        if (threadIdx.x == 0 && (se | be)) {
            atomicAdd(&ws[2 * rl], se);
            atomicAdd(&ws[2 * rl + 1], be);
        }
    }
}

// =================================================================================================
// C4: NA x NA flat MIMO + per-antenna OFDM.  One workgroup per realization; the NA antenna streams
// of one OFDM symbol live in LDS and are transformed together.
// =================================================================================================
struct MimoParams {
    int cp, num_used, n_ofdm_sym;
    int mmse;
    double noise_var;
};

template <typename T, int N, int NA>
__global__ __launch_bounds__(kPipeBlock, sizeof(T) == 4 ? 3 : 1) void k_run_mimo_ofdm(MimoParams pp, ModemParams<T> mp, uint64_t seed,
                                                              uint64_t first, uint64_t count,
                                                              const cx<T>* __restrict__ g_tw,
                                                              cx<T>* g_filters,
                                                              mcle_counters* counters,
                                                              uint32_t* __restrict__ sym_out,
                                                              uint32_t* __restrict__ bit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* s_x = reinterpret_cast<cx<T>*>(smem);        // [NA][N]
    cx<T>* s_tw = s_x + NA * N;                          // [N]
    cx<T>* s_table = s_tw + N;                           // [kMaxTable]
    cx<T>* s_H = s_table + kMaxTable;                    // [NA*NA]
    cx<T>* s_G = s_H + NA * NA;                          // [NA*NA]
    float4* s_tab4 = reinterpret_cast<float4*>(s_G + NA * NA);     // [kMaxTable] {re, im, |c|^2/2, 0} (f32 min-distance)
    unsigned* s_red = reinterpret_cast<unsigned*>(s_tab4 + kMaxTable);  // [8] + flag
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_red + 16);   // [G*G] candidate grid (f32)
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);  // [NA*num_used]

    const int tid = threadIdx.x;
    for (int k = tid; k < N; k += kPipeBlock) s_tw[k] = g_tw[k];
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    for (int m = tid; m < mp.M; m += kPipeBlock) {
        const cx<T> c = mp.g_table[m];
        s_tab4[m] = make_float4((float)c.x, (float)c.y, (float)(0.5 * (c.x * c.x + c.y * c.y)), 0.f);
    }
    const int U = pp.num_used, cp = pp.cp;
    const int per_sym = U * NA;                       // data symbols per OFDM symbol (all antennas)
    const uint64_t row = (uint64_t)pp.n_ofdm_sym * (N + cp);
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)NA) / sqrt((double)(U + cp)));  // encode / sqrt(Nt), ifft power scale
    const double rx_scale = sqrt((double)(U + cp)) / (double)N;               // fft / sqrt(power scale)
    const uint32_t mask = (uint32_t)(mp.M - 1);
    __shared__ WgTotals totals;
    if (threadIdx.x == 0) wg_zero(totals);

    // one butterfly position per thread and stage (N = 4 * threads, radix-4 only): its twiddles live in registers
    constexpr bool kTwRegs = (N == 4 * kPipeBlock) && !FftShape<N>::HAS2 && sizeof(T) == 4;
    cx<T> twr[FftShape<N>::N4][3];
    if constexpr (kTwRegs) {
        __syncthreads();
        fft_twiddle_regs<T, N, kPipeBlock>(s_tw, twr);
    }
    // Channel draw and receive filter, 64 realizations at a time: lane j of wave 0 prepares the j-th of this
    // workgroup's next 64 realizations (the f64 Cholesky is ~750 double-precision instructions -- run by one lane
    // per realization it stalled the other 255 threads for ~9 % of the kernel) and parks H and G in the workgroup's
    // slice of g_filters [gridDim.x][64][2*NA*NA + 1]; the realization loop then stages one record through LDS.
    constexpr int kRec = 2 * NA * NA + 1;
    cx<T>* my_filters = g_filters + (size_t)blockIdx.x * 64 * kRec;
    uint64_t it = 0;
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x, ++it) {
        const Rng rng(seed, first + rl);
        const int slot = (int)(it & 63);
        __syncthreads();
        if (slot == 0) {
            const uint64_t rj = rl + (uint64_t)tid * gridDim.x;
            if (tid < 64 && rj < count) {
                const Rng rngj(seed, first + rj);
                cx<T>* rec = my_filters + tid * kRec;
                double2 H[NA][NA], G[NA][NA];
#pragma unroll
                for (int r = 0; r < NA; ++r)
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        const cx<T> h = cn_sample<T>(rngj, STREAM_CHAN, (uint64_t)(r * NA + a), (T)1);
                        rec[r * NA + a] = h;
                        H[r][a] = mk<double>((double)h.x, (double)h.y);
                    }
                const bool ok = blast_filter<NA, NA>(H, pp.mmse ? pp.noise_var : 0.0, G);
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int r = 0; r < NA; ++r)
                        rec[NA * NA + a * NA + r] = mk<T>((T)(G[a][r].x * rx_scale), (T)(G[a][r].y * rx_scale));
                rec[2 * NA * NA] = mk<T>(ok ? (T)0 : (T)1, (T)0);
            }
            __threadfence_block();
            __syncthreads();
        }
        if (tid < kRec) {                                                      // per-lane addresses: vector loads
            const cx<T> v = my_filters[slot * kRec + tid];
            if (tid < 2 * NA * NA) s_H[tid] = v;                               // s_G follows s_H in LDS
            else s_red[15] = v.x != (T)0 ? 1u : 0u;
        }
        __syncthreads();
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            // ---- transmit: symbols -> bins (Blast.encode's F-order split + OFDM subcarrier map) ----
            if (U != N) {
                for (int p = tid; p < NA * N; p += kPipeBlock) s_x[p] = mk<T>(0, 0);
                __syncthreads();
            }
            const uint64_t n_first = (uint64_t)os * per_sym;
            const uint64_t n_last = n_first + per_sym;
            const bool aligned_scatter = U == N && (per_sym & 15) == 0 && (16 % NA) == 0;
            for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += kPipeBlock) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
                if (aligned_scatter) {   // full band on block boundaries: bins bin(d0) ^ t, the swizzle is XOR-linear
                    const int nl0 = (int)((blk << 4) - n_first);
                    const int p0 = lds_swz<true>(ofdm_bin(nl0 / NA, N, U));
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        s_idx[nl0 + j] = (unsigned char)tx;
                        s_x[(j % NA) * N + (p0 ^ lds_swz<true>(j / NA))] = cscale(s_table[tx], tx_scale);
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
                        s_x[a * N + lds_swz<true>(ofdm_bin(d, N, U))] = cscale(s_table[tx], tx_scale);
                    }
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dif_r<T, N, true, kPipeBlock, true>(s_x, NA, N, twr);
            else
                fft_dif<T, N, true, kPipeBlock, true>(s_x, NA, N, s_tw);  // time samples, digit-reversed positions
            // ---- channel: R = H T + noise on the samples that survive CP removal ----
            cx<T> H[NA][NA];                                          // loaded here: not live across the transform
#pragma unroll
            for (int r = 0; r < NA; ++r)
#pragma unroll
                for (int a = 0; a < NA; ++a) H[r][a] = s_H[r * NA + a];
            for (int j = opaque(tid); j < N / 2; j += kPipeBlock) {   // phase-local addresses (see fft.hpp FRESH)
                const int half = j / (N / 4), rest = j - half * (N / 4);
                const int p0 = 2 * half * (N / 4) + rest, p1 = p0 + N / 4;
                const int m0 = fft_index_of_pos<N>(p0);  // even; position p1 holds m0 + 1
                const int q0 = lds_swz<true>(p0), q1 = lds_swz<true>(p1);
                cx<T> x0[NA], x1[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    x0[a] = s_x[a * N + q0];
                    x1[a] = s_x[a * N + q1];
                }
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const uint64_t i0 = (uint64_t)r * row + (uint64_t)os * (N + cp) + cp + m0;
                    cx<T> z0, z1;
                    if ((i0 & 1) == 0) {
                        cn_pair<T>(rng, STREAM_NOISE, (uint32_t)(i0 >> 1), sigma, z0, z1);
                    } else {
                        z0 = cn_sample<T>(rng, STREAM_NOISE, i0, sigma);
                        z1 = cn_sample<T>(rng, STREAM_NOISE, i0 + 1, sigma);
                    }
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        z0 = cfma4(H[r][a], x0[a], z0);
                        z1 = cfma4(H[r][a], x1[a], z1);
                    }
                    s_x[r * N + q0] = z0;
                    s_x[r * N + q1] = z1;
                }
            }
            __syncthreads();
            if constexpr (kTwRegs)
                fft_dit_r<T, N, false, kPipeBlock, true>(s_x, NA, N, twr);
            else
                fft_dit<T, N, false, kPipeBlock, true>(s_x, NA, N, s_tw);  // bins, natural order
            // ---- receive: Blast decode (G already carries the FFT scale), demodulate, count ----
            cx<T> G[NA][NA];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int r = 0; r < NA; ++r) G[a][r] = s_G[a * NA + r];
            for (int d = opaque(tid); d < U; d += kPipeBlock) {
                const int bin = lds_swz<true>(ofdm_bin(d, N, U));
                cx<T> y[NA];
#pragma unroll
                for (int r = 0; r < NA; ++r) y[r] = s_x[r * N + bin];
                cx<T> est[NA];
                int dec[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    est[a] = mk<T>(0, 0);
#pragma unroll
                    for (int r = 0; r < NA; ++r) est[a] = cfma4(G[a][r], y[r], est[a]);
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
            __syncthreads();
        }
        block_sum2(se, be, s_red);
        if (tid == 0) wg_account(totals, se, be, s_red[15] != 0u, rl, sym_out, bit_out);
    }
    if (tid == 0)
        wg_flush(totals, counters, (unsigned long long)per_sym * pp.n_ofdm_sym,
                 (unsigned long long)per_sym * pp.n_ofdm_sym * mp.bits);
}

// =================================================================================================
// C3: SISO OFDM over a time-varying Jakes TDL channel with the one-tap equaliser.  One workgroup
// (BLOCK threads) per realization.  LDS: current / previous time-domain symbol + receive buffer.
// =================================================================================================
struct TdlParams {
    int cp, num_used, n_ofdm_sym;
    int n_taps, L;
    double noise_var, Fd, Ts, dt;
    double tap_amp[MCLE_MAX_TAPS];   // sqrt(p_i / L)
    int tap_delay[MCLE_MAX_TAPS];
};

template <typename T, int N, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_run_ofdm_tdl(TdlParams pp, ModemParams<T> mp, uint64_t seed,
                                                        uint64_t first, uint64_t count,
                                                        const cx<T>* __restrict__ g_tw, mcle_counters* counters,
                                                        uint32_t* __restrict__ sym_out,
                                                        uint32_t* __restrict__ bit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* s_a = reinterpret_cast<cx<T>*>(smem);   // [N] time-domain symbol (ping)
    cx<T>* s_b = s_a + N;                           // [N] time-domain symbol (pong)
    cx<T>* s_y = s_b + N;                           // [N] received symbol
    cx<T>* s_tw = s_y + N;                          // [N]
    cx<T>* s_table = s_tw + N;                      // [kMaxTable]
    cx<T>* s_mean = s_table + kMaxTable;            // [MCLE_MAX_TAPS]
    cx<T>* s_part = s_mean + MCLE_MAX_TAPS;         // [MCLE_MAX_TAPS][BLOCK/64]
    double* s_w = reinterpret_cast<double*>(s_part + MCLE_MAX_TAPS * (BLOCK / 64));  // [taps*L]
    double* s_psi = s_w + MCLE_MAX_TAPS * kMaxRays / 4;                              // [taps*L]
    double2* s_rayD = reinterpret_cast<double2*>(s_psi + MCLE_MAX_TAPS * kMaxRays / 4);   // [taps*L] ray sums
    float2* s_rotB = reinterpret_cast<float2*>(s_rayD + MCLE_MAX_TAPS * kMaxRays / 4);     // [taps*L] BLOCK-step rotations
    unsigned* s_red = reinterpret_cast<unsigned*>(s_rotB + MCLE_MAX_TAPS * kMaxRays / 4);
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_red + 16);  // [G*G] candidate grid (f32)
    unsigned char* s_idx = reinterpret_cast<unsigned char*>(s_grid + mp.grid.G * mp.grid.G);   // [num_used]

    const int tid = threadIdx.x;
    for (int k = tid; k < N; k += BLOCK) s_tw[k] = g_tw[k];
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    const int U = pp.num_used, cp = pp.cp, S = pp.n_taps, L = pp.L;
    const int dmax = pp.tap_delay[S - 1];
    const T sigma = (T)sqrt(pp.noise_var);
    const T tx_scale = (T)(1.0 / sqrt((double)(U + cp)));
    const T rx_scale = (T)(sqrt((double)(U + cp)) / (double)N);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const int lane = tid & 63, wave = tid >> 6;
    __shared__ WgTotals totals;
    if (threadIdx.x == 0) wg_zero(totals);

    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x) {
        const Rng rng(seed, first + rl);
        __syncthreads();
        // Jakes phases for (L, taps): TdlChannel ctor re-draw, fading.py:796-798
        for (int q = tid; q < S * L; q += BLOCK) {
            const int l = q / S, s = q - l * S;
            const double two_pi = 6.283185307179586476925286766559;
            const double phi = two_pi * uniform_at(rng, STREAM_PHASE, (uint64_t)(l * S + s));
            const double psi = two_pi * uniform_at(rng, STREAM_PHASE, (uint64_t)(L * S + l * S + s));
            if (sizeof(T) == 8) {
                s_w[s * L + l] = two_pi * pp.Fd * cos(phi);
                s_psi[s * L + l] = psi;
            } else {
                const double w = pp.Fd * cos(phi);
                s_w[s * L + l] = w;
                s_psi[s * L + l] = psi / two_pi;
                double rs, rc;
                sincos(two_pi * (w * pp.dt * BLOCK), &rs, &rc);
                s_rotB[s * L + l] = make_float2((float)rc, (float)rs);
            }
        }
        cx<T>* cur = s_a;
        cx<T>* prev = s_b;
        unsigned se = 0, be = 0;
        for (int os = 0; os < pp.n_ofdm_sym; ++os) {
            for (int p = tid; p < N; p += BLOCK) cur[p] = mk<T>(0, 0);
            __syncthreads();
            const uint64_t n_first = (uint64_t)os * U, n_last = n_first + U;
            for (uint64_t blk = (n_first >> 4) + tid; blk <= ((n_last - 1) >> 4); blk += BLOCK) {
                const Words4 dw = rng.block(STREAM_DATA, (uint32_t)blk);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t n = (blk << 4) + j;
                    if (n >= n_first && n < n_last) {
                        const int tx = (int)((dw.w[j >> 2] >> ((j & 3) * 8)) & mask);
                        const int d = (int)(n - n_first);
                        s_idx[d] = (unsigned char)tx;
                        cur[fft_pos_of_index<N>(ofdm_bin(d, N, U))] = cscale(s_table[tx], tx_scale);
                    }
                }
            }
            __syncthreads();
            fft_dit<T, N, true, BLOCK>(cur, 1, N, s_tw);   // bins scattered digit-reversed -> time samples in natural order
            // ---- channel + noise for the N samples kept after CP removal, and the tap means ----
            const uint64_t sym0 = (uint64_t)os * (N + cp);  // absolute index of this symbol's first sample
            // the sample (q within this symbol, or of the previous one) that tap delay d feeds into output m
            auto tx_sample = [&](int m, int d) -> cx<T> {
                const int q = cp + m - d;
                if (q >= 0) return cur[(m - d + N) & (N - 1)];
                const int qp = N + cp + q;  // inter-symbol interference: sample qp of the previous symbol
                return qp >= cp ? prev[qp - cp] : prev[N - cp + qp];
            };
            if constexpr (sizeof(T) == 4) {
                // f32 path.  (1) mean of every ray over the symbol's N+cp samples in closed form (f64):
                //   sum_{j<W} e^{2 pi i (th0 + a j)} = e^{2 pi i (th0 + a (W-1)/2)} sin(pi a W) / sin(pi a)
                const int Wn = N + cp;
                for (int q = tid; q < S * L; q += BLOCK) {
                    const double pi = 3.14159265358979323846;
                    const double w = s_w[q], a = w * pp.dt;
                    const double th0 = fma(w, jakes_time(pp.Ts, pp.dt, (double)sym0), s_psi[q]);
                    const double den = sin(pi * a);
                    const double ratio = fabs(den) > 1e-300 ? sin(pi * a * Wn) / den : (double)Wn;
                    double sn, cs;
                    const double th = th0 + 0.5 * a * (Wn - 1);
                    sincos(2.0 * pi * (th - floor(th)), &sn, &cs);
                    s_rayD[q] = mk<double>(ratio * cs, ratio * sn);
                }
                __syncthreads();
                if (tid < S) {
                    double re = 0, im = 0;
                    for (int l = 0; l < L; ++l) {
                        re += s_rayD[tid * L + l].x;
                        im += s_rayD[tid * L + l].y;
                    }
                    const double a = pp.tap_amp[tid] / (double)Wn;
                    s_mean[tid] = mk<T>((T)(re * a), (T)(im * a));
                }
                // (2) channel: lane handles samples m = tid + BLOCK*k; every ray is evaluated once from
                // the closed form (f64 phase) and stepped by the rotation e^{2 pi i w dt BLOCK}
                constexpr int KS = N / BLOCK;
                cx<T> sig[KS];
#pragma unroll
                for (int k = 0; k < KS; ++k) sig[k] = mk<T>(0, 0);
                for (int i = 0; i < S; ++i) {
                    const int d = pp.tap_delay[i];
                    const long long j0 = (long long)(sym0 + cp + tid) - d;
                    const double t = jakes_time(pp.Ts, pp.dt, (double)j0);
                    cx<T> g[KS];
#pragma unroll
                    for (int k = 0; k < KS; ++k) g[k] = mk<T>(0, 0);
                    for (int l = 0; l < L; ++l) {
                        cx<T> ph = jakes_ray<T>(s_w[i * L + l], s_psi[i * L + l], t);
                        const cx<T> rot = s_rotB[i * L + l];
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            g[k] = cadd(g[k], ph);
                            ph = cmul(ph, rot);
                        }
                    }
                    const T a = (T)pp.tap_amp[i];
#pragma unroll
                    for (int k = 0; k < KS; ++k)
                        if (j0 + (long long)BLOCK * k >= 0)  // before the start of the stream: nothing to add
                            sig[k] = cadd(sig[k], cmul(cscale(g[k], a), tx_sample(tid + BLOCK * k, d)));
                }
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int m = tid + BLOCK * k;
                    s_y[m] = cadd(sig[k], cn_sample<T>(rng, STREAM_NOISE, sym0 + cp + m, sigma));
                }
            } else {
                // f64 parity path: every tap gain from the closed form at its own sample time, the mean
                // as the explicit sum over the symbol's samples (ofdm.py:545-547)
                for (int i = 0; i < S; ++i) {
                    T are = 0, aim = 0;
                    for (int jp = tid; jp < N + cp; jp += BLOCK) {
                        const double t = jakes_time(pp.Ts, pp.dt, (double)(sym0 + jp));
                        T gr = 0, gi = 0;
                        for (int l = 0; l < L; ++l) {
                            const cx<T> ray = jakes_ray<T>(s_w[i * L + l], s_psi[i * L + l], t);
                            gr += ray.x;
                            gi += ray.y;
                        }
                        are += gr;
                        aim += gi;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        are += __shfl_xor(are, off, 64);
                        aim += __shfl_xor(aim, off, 64);
                    }
                    if (lane == 0) s_part[i * (BLOCK / 64) + wave] = mk<T>(are, aim);
                }
                __syncthreads();
                if (tid < S) {
                    T re = 0, im = 0;
                    for (int w = 0; w < BLOCK / 64; ++w) {
                        re += s_part[tid * (BLOCK / 64) + w].x;
                        im += s_part[tid * (BLOCK / 64) + w].y;
                    }
                    const T a = (T)pp.tap_amp[tid] / (T)(N + cp);
                    s_mean[tid] = mk<T>(re * a, im * a);
                }
                for (int m = tid; m < N; m += BLOCK) {
                    const uint64_t jabs = sym0 + cp + m;  // absolute sample index of output sample m
                    const cx<T> acc = cn_sample<T>(rng, STREAM_NOISE, jabs, sigma);
                    cx<T> sig = mk<T>(0, 0);
                    for (int i = 0; i < S; ++i) {
                        const int d = pp.tap_delay[i];
                        const long long jsrc = (long long)jabs - d;  // sample the tap multiplies
                        if (jsrc < 0) continue;                      // before the start of the stream
                        const double t = jakes_time(pp.Ts, pp.dt, (double)jsrc);
                        T gr = 0, gi = 0;
                        for (int l = 0; l < L; ++l) {
                            const cx<T> ray = jakes_ray<T>(s_w[i * L + l], s_psi[i * L + l], t);
                            gr += ray.x;
                            gi += ray.y;
                        }
                        const T a = (T)pp.tap_amp[i];
                        sig = cadd(sig, cmul(mk<T>(a * gr, a * gi), tx_sample(m, d)));
                    }
                    s_y[m] = cadd(sig, acc);
                }
            }
            __syncthreads();
            fft_dif<T, N, false, BLOCK>(s_y, 1, N, s_tw);   // bins, digit-reversed positions
            for (int d = tid; d < U; d += BLOCK) {
                const int bin = ofdm_bin(d, N, U);
                cx<T> h = mk<T>(0, 0);
                for (int i = 0; i < S; ++i) h = cfma4(s_mean[i], s_tw[(bin * pp.tap_delay[i]) & (N - 1)], h);
                const cx<T> eq = cdivide(cscale(s_y[fft_pos_of_index<N>(bin)], rx_scale), h);
                const unsigned x = (unsigned)((int)s_idx[d] ^ demod_one(mp, s_table, s_grid, eq));
                se += (x != 0u);
                be += __popc(x);
            }
            __syncthreads();
            cx<T>* tmp = cur;
            cur = prev;
            prev = tmp;
        }
        (void)dmax;
        block_sum2(se, be, s_red);
        if (tid == 0) wg_account(totals, se, be, false, rl, sym_out, bit_out);
    }
    if (tid == 0)
        wg_flush(totals, counters, (unsigned long long)U * pp.n_ofdm_sym,
                 (unsigned long long)U * pp.n_ofdm_sym * mp.bits);
}

// ---- host side -----------------------------------------------------------------------------------
// workspace = [2*count] error partials + [count] skip flags, zeroed
int pipe_workspace(mcle_ctx* ctx, uint64_t count, unsigned** ws, unsigned** skipped) {
    void* p = nullptr;
    int rc = ctx->scratch((size_t)count * 3 * sizeof(unsigned), &p);
    if (rc) return rc;
    MCLE_HIP(hipMemsetAsync(p, 0, (size_t)count * 3 * sizeof(unsigned), ctx->stream));
    *ws = (unsigned*)p;
    *skipped = (unsigned*)p + 2 * count;
    return MCLE_OK;
}

int pipe_fold(mcle_ctx* ctx, const unsigned* ws, const unsigned* skipped, uint64_t count, uint64_t n_sym,
              mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    hipLaunchKernelGGL(k_fold_counters, dim3(grid_for(ctx, count, 256, 1)), dim3(256), 0, ctx->stream, ws, skipped,
                       (size_t)count, (unsigned long long)n_sym, (unsigned long long)n_sym * ctx->bits, d_counters,
                       d_sym, d_bit);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

template <typename T>
int run_flat_impl(mcle_ctx* ctx, const FlatParams& fp, int method, uint64_t seed, uint64_t first, uint64_t count,
                  mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    unsigned *ws = nullptr, *sk = nullptr;
    int rc = pipe_workspace(ctx, count, &ws, &sk);
    if (rc) return rc;
    const uint64_t items = count * (uint64_t)((fp.n_symbols + kChunk - 1) / kChunk);
    ModemParams<T> mp = pipe_modem<T>(ctx, method);
    const size_t lds = (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long);
    // f32 Jakes links of 8 / 16 rays: the ray sum on the matrix cores (MCLE_OPT_NO_MFMA keeps the VALU recurrence);
    // the f32 kernels are specialised by demodulator (1 packed slicer, 2 lockstep min-distance search, 0 the rest)
    const bool mfma = sizeof(T) == 4 && !fp.rayleigh_iid && (fp.L == 8 || fp.L == 16) && !ctx->opt[MCLE_OPT_NO_MFMA];
    const int mode = sizeof(T) == 8 ? 0
                     : mp.method == MCLE_DEMOD_QAM_SLICER ? 1
                     : (mp.method == MCLE_DEMOD_MINDIST && (mp.M <= 8 || mp.grid.G > 0)) ? 2 : 0;
    const int lr = sizeof(T) == 4 && !fp.rayleigh_iid && (fp.L == 8 || fp.L == 16) ? fp.L : 0;
    // 64 workgroups per CU although 3 - 5 are resident (164 / 80 registers): the queued ones start as the first finish, which
    // evens out per-workgroup speed differences and shortens the tail.  Measured (MCLE_OPT_FLAT_WGS_PER_CU): AWGN 10^4 symbols
    // 5.3 / 5.6 / 5.9 / 6.1 / 6.3 / 6.4 e7 realizations/s at 6 / 8 / 16 / 32 / 64 / 512 per CU, config 2 3.89 / 3.98 / 4.03 /
    // 4.09 / 4.16 / 3.83 e6 (past 64 a workgroup no longer spans a realization's chunks and redoes the ray set-up); a grid
    // of exactly the resident set was 6 - 9 % slower than 8.
    int per_cu = 64;
    if (ctx->opt[MCLE_OPT_FLAT_WGS_PER_CU] > 0) per_cu = (int)ctx->opt[MCLE_OPT_FLAT_WGS_PER_CU];
    const uint64_t cap = (uint64_t)ctx->n_cu * (uint64_t)per_cu;
    const unsigned grid = (unsigned)(items < cap ? items : cap);
#define MCLE_FLAT_LAUNCH(KERN) \
    hipLaunchKernelGGL((KERN), dim3(grid), dim3(kPipeBlock), lds, ctx->stream, fp, mp, seed, first, count, ws)
    if constexpr (sizeof(T) == 4) {
        switch ((mfma ? 100 : 0) + lr * 3 + mode) {
            case 100 + 24 + 0: MCLE_FLAT_LAUNCH((k_run_flat_mfma<8, 0>)); break;
            case 100 + 24 + 1: MCLE_FLAT_LAUNCH((k_run_flat_mfma<8, 1>)); break;
            case 100 + 24 + 2: MCLE_FLAT_LAUNCH((k_run_flat_mfma<8, 2>)); break;
            case 100 + 48 + 0: MCLE_FLAT_LAUNCH((k_run_flat_mfma<16, 0>)); break;
            case 100 + 48 + 1: MCLE_FLAT_LAUNCH((k_run_flat_mfma<16, 1>)); break;
            case 100 + 48 + 2: MCLE_FLAT_LAUNCH((k_run_flat_mfma<16, 2>)); break;
            case 24 + 0: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 0>)); break;
            case 24 + 1: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 1>)); break;
            case 24 + 2: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 2>)); break;
            case 48 + 0: MCLE_FLAT_LAUNCH((k_run_flat<T, 16, 0>)); break;
            case 48 + 1: MCLE_FLAT_LAUNCH((k_run_flat<T, 16, 1>)); break;
            case 48 + 2: MCLE_FLAT_LAUNCH((k_run_flat<T, 16, 2>)); break;
            case 1: MCLE_FLAT_LAUNCH((k_run_flat<T, 0, 1>)); break;
            case 2: MCLE_FLAT_LAUNCH((k_run_flat<T, 0, 2>)); break;
            default: MCLE_FLAT_LAUNCH((k_run_flat<T, 0, 0>)); break;
        }
    } else {
        // complex128: the same rotation recurrence over a thread's 16 symbols (15 complex products after an exact
        // start: <= 3e-15 relative, against 8 or 16 f64 sincos per symbol); MCLE_OPT_JAKES_DIRECT evaluates every sample
        const int lr64 = !fp.rayleigh_iid && (fp.L == 8 || fp.L == 16) && !ctx->opt[MCLE_OPT_JAKES_DIRECT] ? fp.L : 0;
        const int dec = lr64 == 8 ? walk_dec_kind(ctx, mp) : WDEC_GENERIC;     // (fills the on-axis certificate's constants into mp)
        switch (lr64 == 8 ? 80 + dec : lr64) {
            case 80 + WDEC_SLICER: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 10 + WDEC_SLICER>)); break;
            case 80 + WDEC_QAM_CERT: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 10 + WDEC_QAM_CERT>)); break;
            case 80 + WDEC_QUAD_CERT: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 10 + WDEC_QUAD_CERT>)); break;
            case 80 + WDEC_AXIS4_CERT: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 10 + WDEC_AXIS4_CERT>)); break;
            case 80 + WDEC_GENERIC: MCLE_FLAT_LAUNCH((k_run_flat<T, 8, 0>)); break;
            case 16: MCLE_FLAT_LAUNCH((k_run_flat<T, 16, 0>)); break;
            default: MCLE_FLAT_LAUNCH((k_run_flat<T, 0, 0>)); break;
        }
    }
#undef MCLE_FLAT_LAUNCH
    MCLE_LAUNCH_CHECK();
    return pipe_fold(ctx, ws, nullptr, count, (uint64_t)fp.n_symbols, d_counters, d_sym, d_bit);
}

template <typename T, int N, int NA>
int run_mimo_impl(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                  mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    MimoParams pp{cfg->cp_size, cfg->num_used, cfg->n_ofdm_sym, cfg->mmse, cfg->noise_var};
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const size_t lds = (size_t)(NA * N + N + kMaxTable + 2 * NA * NA) * sizeof(cx<T>) + kMaxTable * sizeof(float4) +
                       16 * sizeof(unsigned) + (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       (size_t)NA * cfg->num_used;
    MCLE_REQUIRE(lds <= 160 * 1024, "configuration needs %zu bytes of LDS (limit 160 KiB)", lds);
    auto kern = k_run_mimo_ofdm<T, N, NA>;
    void* filters = nullptr;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));  // gfx950: 160 KiB of LDS per CU
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    const uint64_t cap = (uint64_t)ctx->n_cu * per_cu;
    const unsigned grid = (unsigned)(count < cap ? count : cap);
    if ((rc = ctx->scratch((size_t)grid * 64 * (2 * NA * NA + 1) * sizeof(cx<T>), &filters))) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPipeBlock), lds, ctx->stream, pp, mp, seed, first, count,
                       (const cx<T>*)tw, (cx<T>*)filters, d_counters, d_sym, d_bit);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

template <typename T, int N>
int run_tdl_impl(mcle_ctx* ctx, const mcle_ofdm_tdl_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                 mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    constexpr int BLOCK = N >= 1024 ? 256 : (N >= 256 ? 128 : 64);
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(N, sizeof(T) == 8 ? MCLE_F64 : MCLE_F32, &tw))) return rc;
    TdlParams pp;
    pp.cp = cfg->cp_size;
    pp.num_used = cfg->num_used;
    pp.n_ofdm_sym = cfg->n_ofdm_sym;
    pp.n_taps = cfg->n_taps;
    pp.L = cfg->L;
    pp.noise_var = cfg->noise_var;
    pp.Fd = cfg->Fd;
    pp.Ts = cfg->Ts;
    // numpy.arange(t0, ..., Ts*1.0000000001): delta = fl(fl(t0 + step) - t0), t0 = Ts (fading_generators.py:459-462)
    {
        volatile double step = cfg->Ts * 1.0000000001;
        volatile double nxt = cfg->Ts + step;
        pp.dt = nxt - cfg->Ts;
    }
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) {
        pp.tap_amp[i] = i < cfg->n_taps ? std::sqrt(cfg->tap_power[i]) * std::sqrt(1.0 / (double)cfg->L) : 0.0;
        pp.tap_delay[i] = i < cfg->n_taps ? cfg->tap_delay[i] : 0;
    }
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const size_t lds = (size_t)(4 * N + kMaxTable + MCLE_MAX_TAPS + MCLE_MAX_TAPS * (BLOCK / 64)) * sizeof(cx<T>) +
                       (size_t)(MCLE_MAX_TAPS * kMaxRays / 4) * (2 * sizeof(double) + sizeof(double2) + sizeof(float2)) +
                       16 * sizeof(unsigned) + (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long) +
                       (size_t)cfg->num_used + 16;
    auto kern = k_run_ofdm_tdl<T, N, BLOCK>;
    MCLE_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((size_t)160 * 1024 / (lds + 512));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2048 / BLOCK) per_cu = 2048 / BLOCK;
    const uint64_t cap = (uint64_t)ctx->n_cu * per_cu;
    const unsigned grid = (unsigned)(count < cap ? count : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, ctx->stream, pp, mp, seed, first, count, (const cx<T>*)tw,
                       d_counters, d_sym, d_bit);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // namespace mcle

namespace mcle {
// pipeline_mimo_mfma.hip: f32, FFT 1024, 4x4 on the matrix cores; MCLE_E_UNSUPPORTED outside that envelope
// pipeline_mimo_fw.hip: one realization (4 x 4) / two realizations (2 x 2) per wavefront at fft_size 256, complex128
int run_mimo_ofdm_fw(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                     mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
int run_mimo_ofdm_planar(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                      mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
int run_mimo_ofdm_mfma(mcle_ctx* ctx, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                       mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
// pipeline_siso_tdl.hip
int run_ofdm_tdl_batched(mcle_ctx* ctx, int dtype, const mcle_ofdm_tdl_cfg* cfg, uint64_t seed, uint64_t first,
                         uint64_t count, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit);
}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_run_awgn(mcle_ctx* ctx, int dtype, const mcle_awgn_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                  mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    MCLE_REQUIRE(cfg->n_symbols >= 1, "n_symbols must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    FlatParams fp{cfg->n_symbols, 0, 0, 0.0, 0.0, 0.0, std::sqrt(cfg->noise_var)};
    if (dtype == MCLE_F32)
        return run_flat_impl<float>(ctx, fp, cfg->demod_method, seed, first, count, d_counters, d_sym_err, d_bit_err);
    return run_flat_impl<double>(ctx, fp, cfg->demod_method, seed, first, count, d_counters, d_sym_err, d_bit_err);
}

int mcle_run_flat_fading(mcle_ctx* ctx, int dtype, const mcle_flat_cfg* cfg, uint64_t seed, uint64_t first,
                         uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    MCLE_REQUIRE(cfg->n_symbols >= 1, "n_symbols must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(cfg->rayleigh_iid || (cfg->L >= 1 && cfg->L <= kMaxRays), "L must be in [1, %d]", kMaxRays);
    MCLE_REQUIRE(cfg->rayleigh_iid || cfg->Ts > 0.0, "Ts must be positive");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    FlatParams fp;
    fp.n_symbols = cfg->n_symbols;
    fp.L = cfg->rayleigh_iid ? 0 : cfg->L;
    fp.rayleigh_iid = cfg->rayleigh_iid ? 1 : 0;
    fp.Fd = cfg->Fd;
    fp.t0 = cfg->Ts;  // the Jakes ctor consumed t = 0 (fading_generators.py:348-351)
    {
        volatile double step = cfg->Ts * 1.0000000001;
        volatile double nxt = cfg->Ts + step;
        fp.dt = nxt - cfg->Ts;
    }
    fp.noise_sigma = std::sqrt(cfg->noise_var);
    if (dtype == MCLE_F32)
        return run_flat_impl<float>(ctx, fp, cfg->demod_method, seed, first, count, d_counters, d_sym_err, d_bit_err);
    return run_flat_impl<double>(ctx, fp, cfg->demod_method, seed, first, count, d_counters, d_sym_err, d_bit_err);
}

int mcle_run_mimo_ofdm(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed, uint64_t first,
                       uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    MCLE_REQUIRE(cfg->nt >= 1 && cfg->nt <= cfg->nr && cfg->nr <= 4, "fused MIMO pipeline: 1 <= Nt <= Nr <= 4 (got %d x %d)",
                 cfg->nt, cfg->nr);
    MCLE_REQUIRE(cfg->cp_size >= 0 && cfg->cp_size <= cfg->fft_size,
                 "cp_size must be nonnegative and cannot be greater than fft_size");
    MCLE_REQUIRE(cfg->num_used >= 2 && cfg->num_used % 2 == 0 && cfg->num_used <= cfg->fft_size,
                 "Number of used subcarriers must be a multiple of 2 and at most fft_size");
    MCLE_REQUIRE(cfg->n_ofdm_sym >= 1, "n_ofdm_sym must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "Noise variance must be a non-negative value.");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    if (dtype == MCLE_F32 && ctx->opt[MCLE_OPT_F32_MFMA] && !ctx->opt[MCLE_OPT_NO_MFMA]) {   // matrix-core kernel on request (1024, 4x4)
        rc = run_mimo_ofdm_mfma(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    // the planar kernel family (pipeline_mimo_planar.hip: FFT 256 .. 2048, 1 <= Nt <= Nr <= 4).  no_mfma = 1 keeps complex64 on the
    // round-1 generic kernel below WHERE THAT ONE EXISTS (Nt = Nr in {2, 4}); every other geometry still runs the planar family --
    // the option selects a kernel, it does not shrink the envelope
    const bool generic_has_it = cfg->nt == cfg->nr && (cfg->nt == 2 || cfg->nt == 4);
    // 2x2 at 256 points: the generic kernel (four realizations per 256-thread workgroup) is the faster one -- 1.65 against 1.06e8
    // realizations/s in complex64, 1.01 against 0.98e8 in complex128 (profiles/r05/f32_family_rates.json, f64_family_rates.json)
    const bool generic_is_faster = cfg->fft_size == 256 && cfg->nt == 2 && cfg->nr == 2;
    if (generic_is_faster && !(dtype == MCLE_F32 && ctx->opt[MCLE_OPT_NO_MFMA])) {
        // ... and since round 6 the full-wave kernel (pipeline_mimo_fw.hip, two realizations per wavefront) beats both inside its envelope
        const long long thr = ctx->opt[MCLE_OPT_F64_THREADS];
        if (thr == 0 || thr == 260 || thr == 262) {
            rc = run_mimo_ofdm_fw(ctx, dtype, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
            if (rc != MCLE_E_UNSUPPORTED) return rc;
        }
    }
    if (!(dtype == MCLE_F32 && ctx->opt[MCLE_OPT_NO_MFMA] && generic_has_it) && !generic_is_faster) {
        rc = run_mimo_ofdm_planar(ctx, dtype, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    MCLE_REQUIRE(cfg->nt == cfg->nr && (cfg->nt == 2 || cfg->nt == 4),
                 "fused MIMO pipeline: %d x %d at fft_size %d is outside the envelope (2x2 / 4x4 at 64 .. 2048 in both arithmetics; "
                 "every 1 <= Nt <= Nr <= 4 at 256 .. 2048)", cfg->nt, cfg->nr, cfg->fft_size);
#define MCLE_RUN(N_, NA_)                                                                                         \
    if (cfg->fft_size == N_ && cfg->nt == NA_)                                                                    \
        return dtype == MCLE_F32                                                                                  \
                   ? run_mimo_impl<float, N_, NA_>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err) \
                   : run_mimo_impl<double, N_, NA_>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
    MCLE_RUN(64, 2) MCLE_RUN(64, 4) MCLE_RUN(128, 2) MCLE_RUN(128, 4) MCLE_RUN(256, 2) MCLE_RUN(256, 4)
    MCLE_RUN(512, 2) MCLE_RUN(512, 4) MCLE_RUN(1024, 2) MCLE_RUN(1024, 4) MCLE_RUN(2048, 2) MCLE_RUN(2048, 4)
#undef MCLE_RUN
    set_error("fused MIMO pipeline supports fft_size in {64, 128, ..., 2048} (got %d)", cfg->fft_size);
    return MCLE_E_INVAL;
}

int mcle_run_ofdm_tdl(mcle_ctx* ctx, int dtype, const mcle_ofdm_tdl_cfg* cfg, uint64_t seed, uint64_t first,
                      uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    MCLE_REQUIRE(cfg->cp_size >= 0 && cfg->cp_size <= cfg->fft_size,
                 "cp_size must be nonnegative and cannot be greater than fft_size");
    MCLE_REQUIRE(cfg->num_used >= 2 && cfg->num_used % 2 == 0 && cfg->num_used <= cfg->fft_size,
                 "Number of used subcarriers must be a multiple of 2 and at most fft_size");
    MCLE_REQUIRE(cfg->n_ofdm_sym >= 1, "n_ofdm_sym must be positive");
    MCLE_REQUIRE(cfg->n_taps >= 1 && cfg->n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    MCLE_REQUIRE(cfg->L >= 1 && cfg->n_taps * cfg->L <= MCLE_MAX_TAPS * kMaxRays / 4, "too many rays (taps * L <= %d)",
                 MCLE_MAX_TAPS * kMaxRays / 4);
    MCLE_REQUIRE(cfg->Ts > 0.0 && cfg->noise_var >= 0.0, "Ts must be positive and noise_var non-negative");
    for (int i = 0; i < cfg->n_taps; ++i) {
        MCLE_REQUIRE(cfg->tap_delay[i] >= 0 && (i == 0 || cfg->tap_delay[i] > cfg->tap_delay[i - 1]),
                     "tap delays must be non-negative and strictly increasing");
        MCLE_REQUIRE(cfg->tap_delay[i] <= cfg->fft_size, "tap delay beyond one OFDM symbol is not supported");
    }
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    // four realizations per workgroup pass with polynomial taps (pipeline_siso_tdl.hip) where that kernel's
    // envelope allows (FFT 64 / 256 / 1024, moderate Doppler); MCLE_OPT_SINGLE_TDL forces the kernel below
    if (!ctx->opt[MCLE_OPT_SINGLE_TDL]) {
        rc = run_ofdm_tdl_batched(ctx, dtype, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
#define MCLE_RUN(N_)                                                                                        \
    if (cfg->fft_size == N_)                                                                                \
        return dtype == MCLE_F32                                                                            \
                   ? run_tdl_impl<float, N_>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err) \
                   : run_tdl_impl<double, N_>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err);
    MCLE_RUN(64) MCLE_RUN(128) MCLE_RUN(256) MCLE_RUN(512) MCLE_RUN(1024) MCLE_RUN(2048)
#undef MCLE_RUN
    set_error("fused OFDM/TDL pipeline supports fft_size in {64, ..., 2048} (got %d)", cfg->fft_size);
    return MCLE_E_INVAL;
}

}  // extern "C"
