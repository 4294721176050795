// pipeline_mimo_flat.hip -- fused pipeline of the reference's MIMO application (apps/mimo/simulate_mimo.py:68-142):
// per realization a flat channel H = randn_c(Nr, Nt), one of the six schemes
//   Blast (mimo/mimo.py:463-660), MRC (:789-830), MRT (:666-783), Alamouti (:1073-1287), SVDMimo (:833-946),
//   GMDMimo (:952-1067, util/misc.py:18-159)
// NSymbs symbols per layer on a single carrier, Y = H X + sqrt(noise_var) N, decode, demodulate, count.
//
// One wavefront per chunk of 64 realizations (the layout of kernels_ia.hip): phase 1, lane i draws the channel of
// realization i and builds the scheme's precoder W [Nt][layers] and receive filter G [layers][Nr] in f64 registers
// (Cholesky / one-sided Jacobi SVD / GMD rotations) and parks H, W, G in LDS; phase 2, the whole wave runs each
// realization's symbol columns.
// Draw ledger (mcle-philox-v1): CHAN sample r*Nt + a = H[r][a]; DATA symbol n = the reference's flat index
// (Blast / MRC: n = t*Nt + a, Fortran order; SVD / GMD: n = a*NSymbs + t, C order; MRT / Alamouti: n = t);
// NOISE sample r*NSymbs + t.
#include "mimo_svd.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "pipe_common.hpp"
#include "totals.hpp"
#include "qam_pack.hpp"
#include "wave_draws.hpp"

namespace mcle {

constexpr int kFlatMax = 4;   // antennas per side

struct FlatSetup {
    double2 H[kFlatMax][kFlatMax];   // [r][a]
    double2 W[kFlatMax][kFlatMax];   // [a][l]
    double2 G[kFlatMax][kFlatMax];   // [l][r]
    double aux;                      // Alamouti: sqrt(2) / |H|_F^2
    bool ok;
};

template <int NT, int NR>
__device__ __forceinline__ bool flat_blast(const FlatSetup& s, double nv, double2 (&Gout)[kFlatMax][kFlatMax]) {
    double2 Hs[NR][NT], Gs[NT][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int a = 0; a < NT; ++a) Hs[r][a] = s.H[r][a];
    const bool ok = blast_filter<NT, NR>(Hs, nv, Gs);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < NR; ++r) Gout[a][r] = Gs[a][r];
    return ok;
}
template <int NA> __device__ __forceinline__ void flat_svd(FlatSetup& s) {
    double2 A[NA][NA], W[NA][NA], G[NA][NA];
    double S[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) A[r][c] = s.H[r][c];
    svd_filters_dev<NA>(A, W, G, S);
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            s.W[r][c] = W[r][c];
            s.G[r][c] = G[r][c];
        }
}
template <int NA> __device__ __forceinline__ void flat_gmd(FlatSetup& s, double nv) {
    double2 Hs[NA][NA], W[NA][NA], G[NA][NA];
    double R[NA][NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) Hs[r][c] = s.H[r][c];
    s.ok = gmd_filters_dev<NA>(Hs, nv, W, G, R);
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            s.W[r][c] = W[r][c];
            s.G[r][c] = G[r][c];
        }
}

// precoder and receive filter of one realization; not inlined: one compiled body for both instantiations
__device__ __noinline__ void flat_setup(int scheme, int nt, int nr, double nv, FlatSetup& s) {
    const double2 zero = mk<double>(0, 0);
#pragma unroll
    for (int i = 0; i < kFlatMax; ++i)
#pragma unroll
        for (int j = 0; j < kFlatMax; ++j) s.W[i][j] = s.G[i][j] = zero;
    s.aux = 0.0;
    s.ok = true;
    if (scheme == MCLE_MIMO_BLAST || scheme == MCLE_MIMO_MRC) {
        const double w = 1.0 / sqrt((double)nt);
#pragma unroll
        for (int a = 0; a < kFlatMax; ++a) s.W[a][a] = mk<double>(a < nt ? w : 0.0, 0.0);
#define MCLE_FB(NT_, NR_) \
    if (nt == NT_ && nr == NR_) s.ok = flat_blast<NT_, NR_>(s, nv, s.G);
        MCLE_FB(1, 1) MCLE_FB(1, 2) MCLE_FB(1, 3) MCLE_FB(1, 4) MCLE_FB(2, 2) MCLE_FB(2, 3) MCLE_FB(2, 4) MCLE_FB(3, 3)
        MCLE_FB(3, 4) MCLE_FB(4, 4)
#undef MCLE_FB
    } else if (scheme == MCLE_MIMO_MRT) {
        double sum = 0.0;
        const double w = 1.0 / sqrt((double)nt);
#pragma unroll
        for (int a = 0; a < kFlatMax; ++a)
            if (a < nt) {
                const double m = sqrt(s.H[0][a].x * s.H[0][a].x + s.H[0][a].y * s.H[0][a].y);
                sum += m;
                s.W[a][0] = mk<double>(s.H[0][a].x / m * w, -s.H[0][a].y / m * w);   // exp(-j angle(h)) / sqrt(Nt)
            }
        s.G[0][0] = mk<double>(sqrt((double)nt) / sum, 0.0);
    } else if (scheme == MCLE_MIMO_ALAMOUTI) {
        double f = 0.0;
#pragma unroll
        for (int r = 0; r < kFlatMax; ++r)
            if (r < nr) f += s.H[r][0].x * s.H[r][0].x + s.H[r][0].y * s.H[r][0].y + s.H[r][1].x * s.H[r][1].x +
                             s.H[r][1].y * s.H[r][1].y;
        s.aux = sqrt(2.0) / f;
    } else if (scheme == MCLE_MIMO_SVD) {
        if (nt == 2) flat_svd<2>(s);
        else if (nt == 3) flat_svd<3>(s);
        else flat_svd<4>(s);
    } else {
        if (nt == 2) flat_gmd<2>(s, nv);
        else if (nt == 3) flat_gmd<3>(s, nv);
        else flat_gmd<4>(s, nv);
    }
}

// Two launches since round 2 (the split of kernels_ia.hip): the per-lane set-up (Cholesky / Jacobi SVD / GMD in f64)
// takes every register of the SIMD, the symbol walk wants many resident waves.  Record per realization:
// A[4][4], G[4][4], {aux, ok}.
constexpr int kFlatRec = 2 * kFlatMax * kFlatMax + 1;

template <typename T>
__global__ __launch_bounds__(64) void k_mimo_flat_setup(int scheme, int nt, int nr, double filter_nv, uint64_t seed,
                                                        uint64_t first, uint64_t count, cx<T>* __restrict__ recs) {
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    cx<T>* rec = recs + rl * kFlatRec;
    const Rng rng(seed, first + rl);
    FlatSetup st;
#pragma unroll
    for (int r = 0; r < kFlatMax; ++r)
#pragma unroll
        for (int a = 0; a < kFlatMax; ++a)
            st.H[r][a] = (r < nr && a < nt) ? cn_sample<double>(rng, STREAM_CHAN, (uint64_t)(r * nt + a), 1.0)
                                            : mk<double>(0, 0);
    flat_setup(scheme, nt, nr, filter_nv, st);
    if (scheme == MCLE_MIMO_ALAMOUTI) {
#pragma unroll
        for (int i = 0; i < kFlatMax; ++i)
#pragma unroll
            for (int j = 0; j < kFlatMax; ++j)
                rec[i * kFlatMax + j] = mk<T>((T)st.H[i][j].x, (T)st.H[i][j].y);
    } else {
        // A = G (H W) in f64; rows / columns beyond the layers are zero (W, G are zero there)
        double2 HW[kFlatMax][kFlatMax];
#pragma unroll
        for (int r = 0; r < kFlatMax; ++r)
#pragma unroll
            for (int l = 0; l < kFlatMax; ++l) {
                double2 acc = mk<double>(0, 0);
#pragma unroll
                for (int a = 0; a < kFlatMax; ++a) acc = cadd(acc, cmul(st.H[r][a], st.W[a][l]));
                HW[r][l] = acc;
            }
#pragma unroll
        for (int i = 0; i < kFlatMax; ++i)
#pragma unroll
            for (int l = 0; l < kFlatMax; ++l) {
                double2 acc = mk<double>(0, 0);
#pragma unroll
                for (int r = 0; r < kFlatMax; ++r) acc = cadd(acc, cmul(st.G[i][r], HW[r][l]));
                rec[i * kFlatMax + l] = mk<T>((T)acc.x, (T)acc.y);
                rec[kFlatMax * kFlatMax + i * kFlatMax + l] = mk<T>((T)st.G[i][l].x, (T)st.G[i][l].y);
            }
    }
    rec[2 * kFlatMax * kFlatMax] = mk<T>((T)st.aux, st.ok ? (T)1 : (T)0);
}

// acc + a b in the symbol walk: complex64 as before (cfma(float2): four chained FMAs), complex128 chained as well (cfma4; the generic
// cfma keeps the product-then-add association, six operations)
template <typename C> __device__ __forceinline__ C mac(C a, C b, C acc) {
    if constexpr (sizeof(a.x) == 8) return cfma4(a, b, acc);
    else return cfma(a, b, acc);
}

template <typename T>
// (round 6: a workgroup is FOUR independent wavefronts sharing the tables and one flush of the counters -- totals.hpp: wg_flush_waves)
__global__ __launch_bounds__(256, sizeof(T) == 4 ? MCLE_F32_WALK_WAVES : 2) void k_mimo_flat_link(
    ModemParams<T> mp, int scheme, int nt, int nr, int n_symbols, double noise_var, uint64_t seed, uint64_t first,
    uint64_t count, int per_wave, const cx<T>* __restrict__ recs, mcle_counters* counters,
    uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    __shared__ cx<T> s_table[256];
    __shared__ float4 s_tab4[sizeof(T) == 4 ? 256 : 1];     // {re, im, |c|^2 / 2, 0}: the lockstep searches of modem.hpp
    extern __shared__ unsigned long long s_grid[];       // [G*G] candidate grid (min-distance demodulation, f32)
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];   // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, (int)blockDim.x);
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    if constexpr (sizeof(T) == 4)
        for (int m = threadIdx.x; m < mp.M; m += blockDim.x) {
            const float2 c = mp.g_table[m];
            s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
        }
    const bool lockstep = sizeof(T) == 4 && mp.method == MCLE_DEMOD_MINDIST && (mp.M <= 8 || mp.grid.G > 0);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const T sigma = (T)sqrt(noise_var);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const int layers = (scheme == MCLE_MIMO_ALAMOUTI || scheme == MCLE_MIMO_MRT) ? 1 : nt;
    const bool c_order = scheme == MCLE_MIMO_SVD || scheme == MCLE_MIMO_GMD;
    const bool packed = sizeof(T) == 4 && mp.method == MCLE_DEMOD_QAM_SLICER;
    const uint32_t layer_mask = layers >= 4 ? 0xFFFFFFFFu : ((1u << (8 * layers)) - 1u);
    QamPack qp{};
    if constexpr (sizeof(T) == 4) {
        if (packed) qp = qam_pack(mp);
    }
    __shared__ WgTotals totals_all[4];
    WgTotals& totals = totals_all[wv];
    if (lane == 0) wg_zero(totals);
    __syncthreads();
    const uint64_t n_chunks = (count + per_wave - 1) / per_wave;
    for (uint64_t ch = (uint64_t)blockIdx.x * 4 + wv; ch < n_chunks; ch += (uint64_t)gridDim.x * 4) {
        const uint64_t r_end = (ch + 1) * per_wave < count ? (ch + 1) * per_wave : count;
        for (uint64_t rl = ch * per_wave; rl < r_end; ++rl) {
            const Rng rng(seed, first + rl);
            const cx<T>* rec = recs + rl * kFlatRec;     // wave-uniform: scalar loads
            cx<T> A[kFlatMax][kFlatMax], G[kFlatMax][kFlatMax];     // Alamouti: A holds H
#pragma unroll
            for (int i = 0; i < kFlatMax; ++i)
#pragma unroll
                for (int c = 0; c < kFlatMax; ++c) {
                    A[i][c] = rec[i * kFlatMax + c];
                    G[i][c] = rec[kFlatMax * kFlatMax + i * kFlatMax + c];
                }
            unsigned se = 0, be = 0;
            if (scheme == MCLE_MIMO_ALAMOUTI) {
                const T scale = rec[2 * kFlatMax * kFlatMax].x;
                const T inv_root2 = (T)0.70710678118654752440;
                // slot pair p = symbols 2p, 2p+1 (one word of a DATA block, blocks shared across the wave) and the
                // noise samples 2p, 2p+1 of every receive row (one NOISE block each); n_symbols is even
                for (int p0 = 0; p0 < n_symbols / 2; p0 += 64) {
                    const int p = p0 + lane;
                    int ta[1], tb[1];
                    wave_symbol_pairs<1>(rng, 1, 0u, (uint32_t)(2 * p0), mask, lane, ta, tb);
                    if (p < n_symbols / 2) {
                        const int tx0 = ta[0], tx1 = tb[0];
                        const cx<T> s0 = cscale(s_table[tx0], inv_root2), s1 = cscale(s_table[tx1], inv_root2);
                        cx<T> o0 = mk<T>(0, 0), o1 = mk<T>(0, 0);
#pragma unroll
                        for (int r = 0; r < kFlatMax; ++r)
                            if (r < nr) {
                                // slot 2p: (s0, s1); slot 2p+1: (-conj s1, conj s0)
                                cx<T> y0, y1;
                                cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)r * (uint32_t)n_symbols + 2u * (uint32_t)p) >> 1, sigma,
                                            y0, y1, s_bm);
                                y0 = mac(A[r][0], s0, y0);
                                y0 = mac(A[r][1], s1, y0);
                                y1 = mac(A[r][0], mk<T>(-s1.x, s1.y), y1);
                                y1 = mac(A[r][1], cconj(s0), y1);
                                o0 = mac(cconj(A[r][0]), y0, o0);
                                o0 = mac(A[r][1], cconj(y1), o0);
                                o1 = mac(cconj(A[r][1]), y0, o1);
                                o1 = csub(o1, cmul(A[r][0], cconj(y1)));
                            }
                        const unsigned x0 = (unsigned)(tx0 ^ demod_one(mp, s_table, s_grid, cscale(o0, scale)));
                        const unsigned x1 = (unsigned)(tx1 ^ demod_one(mp, s_table, s_grid, cscale(o1, scale)));
                        se += (x0 != 0u) + (x1 != 0u);
                        be += __popc(x0) + __popc(x1);
                    }
                }
            } else {
                // one symbol column: est = A d + G n
                auto column = [&](const int (&tx)[kFlatMax], const cx<T> (&nz)[kFlatMax]) {
                    cx<T> d[kFlatMax];
#pragma unroll
                    for (int l = 0; l < kFlatMax; ++l) d[l] = l < layers ? s_table[tx[l]] : mk<T>(0, 0);
                    cx<T> est[kFlatMax];
#pragma unroll
                    for (int l = 0; l < kFlatMax; ++l) {
                        est[l] = mk<T>(0, 0);
                        if (l < layers) {
#pragma unroll
                            for (int c = 0; c < kFlatMax; ++c) {
                                est[l] = mac(A[l][c], d[c], est[l]);      // entries beyond the layers / rows are zero
                                est[l] = mac(G[l][c], nz[c], est[l]);
                            }
                        }
                    }
                    if constexpr (sizeof(T) == 4) {
                        if (packed) {   // the column's decisions in one packed level-domain slice (qam_pack.hpp)
                            const f4q er = {est[0].x, est[1].x, est[2].x, est[3].x}, ei = {est[0].y, est[1].y, est[2].y, est[3].y};
                            const uint32_t sent = (uint32_t)tx[0] | ((uint32_t)tx[1] << 8) | ((uint32_t)tx[2] << 16) |
                                                  ((uint32_t)tx[3] << 24);
                            qam_count4((qam_levels4(er, ei, qp) ^ labels_to_levels(sent, qp)) & layer_mask, qp, se, be);
                            return;
                        }
                    }
                    int dec[kFlatMax];
                    bool done = false;
                    if constexpr (sizeof(T) == 4) {
                        if (lockstep) {   // the layers searched in lockstep; unused layers carry est = 0 and are not counted
                            if (mp.M <= 8) demod_multi_cert(mp, est, dec, [&](int (&d_)[kFlatMax]) { demod_mindist_multi<kFlatMax>(s_tab4, mp.M, est, d_); });
                            else demod_multi_cert(mp, est, dec, [&](int (&d_)[kFlatMax]) { demod_grid4_multi<kFlatMax>(s_tab4, s_grid, mp.grid, mp.M, est, d_); });
                            done = true;
                        }
                    }
                    if (!done) {
#pragma unroll
                        for (int l = 0; l < kFlatMax; ++l) dec[l] = l < layers ? demod_one(mp, s_table, s_grid, est[l]) : 0;
                    }
#pragma unroll
                    for (int l = 0; l < kFlatMax; ++l)
                        if (l < layers) {
                            const unsigned xr = (unsigned)(tx[l] ^ dec[l]);
                            se += (xr != 0u);
                            be += __popc(xr);
                        }
                };
                if ((n_symbols & 1) == 0) {
                    for (int t0 = 0; t0 < n_symbols; t0 += kPairCols) {
                        const int t = t0 + 2 * lane;
                        int ta[kFlatMax], tb[kFlatMax];
#pragma unroll
                        for (int l = 0; l < kFlatMax; ++l) ta[l] = tb[l] = 0;
                        if (c_order) {
                            wave_symbol_pairs<kFlatMax>(rng, layers, (uint32_t)n_symbols, (uint32_t)t0, mask, lane, ta, tb);
                        } else if (t < n_symbols) {
                            // Fortran order: the 2 * layers bytes of columns t, t + 1 are consecutive (<= 8 bytes,
                            // in one block unless layers == 3)
                            const uint32_t q = (uint32_t)t * (uint32_t)layers;
                            const Words4 b0 = rng.block(STREAM_DATA, q >> 4);
                            Words4 b1 = b0;
                            if ((q & 15u) + 2u * (uint32_t)layers > 16u) b1 = rng.block(STREAM_DATA, (q >> 4) + 1u);
#pragma unroll
                            for (int e = 0; e < 2 * kFlatMax; ++e)
                                if (e < 2 * layers) {
                                    const uint32_t pos = (q & 15u) + (uint32_t)e;          // 0 .. 31
                                    const uint32_t wi = (pos >> 2) & 3u;
                                    const uint32_t lo = wi == 0 ? b0.w[0] : (wi == 1 ? b0.w[1] : (wi == 2 ? b0.w[2] : b0.w[3]));
                                    const uint32_t hi = wi == 0 ? b1.w[0] : (wi == 1 ? b1.w[1] : (wi == 2 ? b1.w[2] : b1.w[3]));
                                    const int v = (int)(((pos < 16u ? lo : hi) >> ((pos & 3u) * 8u)) & mask);
                                    // e = column * layers + layer
#pragma unroll
                                    for (int l = 0; l < kFlatMax; ++l) {
                                        if (e == l) ta[l] = v;
                                        if (e == layers + l) tb[l] = v;
                                    }
                                }
                        }
                        if (t < n_symbols) {
                            cx<T> za[kFlatMax], zb[kFlatMax];
#pragma unroll
                            for (int r = 0; r < kFlatMax; ++r) {
                                za[r] = zb[r] = mk<T>(0, 0);
                                if (r < nr)
                                    cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)r * (uint32_t)n_symbols + (uint32_t)t) >> 1, sigma,
                                                za[r], zb[r], s_bm);
                            }
                            column(ta, za);
                            column(tb, zb);
                        }
                    }
                } else {
                    for (int t = lane; t < n_symbols; t += 64) {
                        int tx[kFlatMax];
                        cx<T> nz[kFlatMax];
#pragma unroll
                        for (int l = 0; l < kFlatMax; ++l) {
                            tx[l] = 0;
                            if (l < layers) {
                                const uint64_t n = c_order ? (uint64_t)l * n_symbols + t : (uint64_t)t * layers + l;
                                tx[l] = (int)symbol_at(rng, n, mask);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < kFlatMax; ++r)
                            nz[r] = r < nr ? cn_sample<T>(rng, STREAM_NOISE, (uint64_t)r * n_symbols + t, sigma) : mk<T>(0, 0);
                        column(tx, nz);
                    }
                }
            }
            se = wave_sum_u32(se);
            be = wave_sum_u32(be);
            if (lane == 0) wg_account(totals, se, be, rec[2 * kFlatMax * kFlatMax].y == (T)0, rl, sym_out, bit_out);
        }
    }
    wg_flush_waves<4>(totals_all, counters, (unsigned long long)layers * n_symbols, (unsigned long long)layers * n_symbols * mp.bits);
}

}  // namespace mcle

using namespace mcle;

extern "C" int mcle_run_mimo_flat(mcle_ctx* ctx, int dtype, const mcle_mimo_flat_cfg* cfg, uint64_t seed, uint64_t first,
                                  uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err,
                                  uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    const int nt = cfg->nt, nr = cfg->nr;
    MCLE_REQUIRE(cfg->scheme >= MCLE_MIMO_BLAST && cfg->scheme <= MCLE_MIMO_GMD, "unknown MIMO scheme %d", cfg->scheme);
    MCLE_REQUIRE(nt >= 1 && nt <= kFlatMax && nr >= 1 && nr <= kFlatMax, "antenna counts must be in [1, %d]", kFlatMax);
    MCLE_REQUIRE(cfg->n_symbols >= 1, "n_symbols must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "Noise variance must be a non-negative value.");
    switch (cfg->scheme) {
        case MCLE_MIMO_BLAST:
            MCLE_REQUIRE(nt <= nr, "the fused Blast pipeline needs Nt <= Nr (got %dx%d)", nr, nt);
            break;
        case MCLE_MIMO_MRC:
            MCLE_REQUIRE(nt == 1, "MRC is the single-transmit-antenna case of Blast (Nt = 1)");
            break;
        case MCLE_MIMO_MRT:   // MisoBase.set_channel_matrix, mimo.py:418-441
            MCLE_REQUIRE(nr == 1, "MISO schemes are only defined for a single receive antenna");
            break;
        case MCLE_MIMO_ALAMOUTI:   // Alamouti.set_channel_matrix, mimo.py:1110-1131
            MCLE_REQUIRE(nt == 2, "The number of transmit antennas must be equal to 2 for the Alamouti scheme");
            MCLE_REQUIRE(cfg->n_symbols % 2 == 0, "Alamouti needs an even number of symbols");
            break;
        default:
            MCLE_REQUIRE(nt == nr && nt >= 2, "SVD / GMD filters support square channels with 2 <= N <= 4");
    }
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    const double filter_nv = cfg->mmse ? cfg->noise_var : 0.0;
    const int per_wave = 16;
    const ModemParams<float> mp32 = pipe_modem<float>(ctx, cfg->demod_method);
    const size_t lds = (size_t)mp32.grid.G * mp32.grid.G * sizeof(unsigned long long);
    const uint64_t kSlice = 1ull << 20;          // realizations per set-up + walk pair: bounds the record buffer
    const uint64_t slice = count < kSlice ? count : kSlice;
    void* recs = nullptr;
    if ((rc = ctx->scratch((size_t)slice * kFlatRec * (dtype == MCLE_F32 ? sizeof(float2) : sizeof(double2)), &recs)))
        return rc;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t m = count - off < slice ? count - off : slice;
        const uint64_t chunks = (m + per_wave - 1) / per_wave;
        const unsigned sgrid = (unsigned)((m + 63) / 64);
        uint32_t* se = d_sym_err ? d_sym_err + off : nullptr;
        uint32_t* be = d_bit_err ? d_bit_err + off : nullptr;
        if (dtype == MCLE_F32) {
            const uint64_t cap = (uint64_t)ctx->n_cu * 16;
            hipLaunchKernelGGL(k_mimo_flat_setup<float>, dim3(sgrid), dim3(64), 0, ctx->stream, cfg->scheme, nt, nr, filter_nv,
                               seed, first + off, m, (float2*)recs);
            MCLE_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_mimo_flat_link<float>, dim3((unsigned)((chunks + 3) / 4 < cap / 4 ? (chunks + 3) / 4 : cap / 4)), dim3(256), lds,
                               ctx->stream, mp32, cfg->scheme, nt, nr, cfg->n_symbols, cfg->noise_var, seed, first + off, m,
                               per_wave, (const float2*)recs, d_counters, se, be);
        } else {
            const uint64_t cap = (uint64_t)ctx->n_cu * 8;
            hipLaunchKernelGGL(k_mimo_flat_setup<double>, dim3(sgrid), dim3(64), 0, ctx->stream, cfg->scheme, nt, nr, filter_nv,
                               seed, first + off, m, (double2*)recs);
            MCLE_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_mimo_flat_link<double>, dim3((unsigned)((chunks + 3) / 4 < cap / 4 ? (chunks + 3) / 4 : cap / 4)), dim3(256), lds,
                               ctx->stream, pipe_modem<double>(ctx, cfg->demod_method), cfg->scheme, nt, nr, cfg->n_symbols,
                               cfg->noise_var, seed, first + off, m, per_wave, (const double2*)recs, d_counters, se, be);
        }
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}
