// kernels_multiuser.hip -- covariance matrices and post-filter SINRs of the K-user interference channel
// (what MultiUserChannelMatrix and MultiUserChannelMatrixExtInt offer around
// `big_H @ X + noise`).
//
// Reference: channels/multiuser.py
//   path loss on the block matrix                      :860-933 (_from_small_matrix_to_big_matrix), :1264-1312, :2415-2467
//   calc_Q / calc_JP_Q                                 :1314-1450, with external interference :2530-2634
//   _calc_Bkl_cov_matrix_{first_part, second_part, all_l} and the JP forms   :1452-1826, :2676-2742
//   _calc_SINR_k / calc_SINR / calc_JP_SINR            :1828-2008, :2636-2674, :2744-2807
//   calc_cov_matrix_extint_{without,plus}_noise        :2469-2520
//
// One thread per (channel realization, receiver k), f64 -- these are a handful of <= 4 x 24 products per call; the
// batch dimension is what fills the device when a simulator evaluates many channels at once.
#include "common.hpp"

namespace mcle {

constexpr int kMuMaxK = 4;       // users
constexpr int kMuMaxAnt = 4;     // antennas per node, streams per user
constexpr int kMuMaxTx = 16;     // sum of transmit antennas
constexpr int kMuMaxExt = 8;     // external interferer antennas

struct MuParams {
    int K, n_ext, joint;
    int nr[kMuMaxK], nt[kMuMaxK], ns[kMuMaxK];
    int sum_nr, sum_nt;
    double noise_var, pe;
};

typedef double2 cd;
__device__ __forceinline__ cd mu_mac(cd acc, cd a, cd b) {          // acc + a b
    acc.x = fma(a.x, b.x, fma(-a.y, b.y, acc.x));
    acc.y = fma(a.x, b.y, fma(a.y, b.x, acc.y));
    return acc;
}
__device__ __forceinline__ cd mu_macc(cd acc, cd a, cd b) {         // acc + a conj(b)
    acc.x = fma(a.x, b.x, fma(a.y, b.y, acc.x));
    acc.y = fma(a.y, b.x, fma(-a.x, b.y, acc.y));
    return acc;
}

// d_bigH [batch][sum_nr][sum_nt + n_ext]; d_pl (may be null) [sum_nr][sum_nt + n_ext] LINEAR path loss per entry
// (shared by the batch); d_F [batch][K][16][4] (rows: nt[j], or sum_nt in joint-processing mode; cols: ns[j]);
// d_U [batch][K][4][4] (nr[k] x ns[k]; may be null -> no SINRs).
// d_Q [batch][K][4][4]: sum_{j != k} G_j G_j^H + Re_k;  d_Re [batch][K][4][4]: pe ext ext^H + noise_var I;
// d_B [batch][K][4 streams][4][4]: (sum_j G_j G_j^H + Re_k) - g_kl g_kl^H;  d_sinr [batch][K][4].
// G_j = H_kj F_j (per-link mode) or H_k F_j with H_k = row block k without the external columns (joint mode).
__global__ __launch_bounds__(64) void k_mu_link_stats(MuParams pp, const cd* __restrict__ d_bigH,
                                                      const double* __restrict__ d_pl, const cd* __restrict__ d_F,
                                                      const cd* __restrict__ d_U, cd* __restrict__ d_Q,
                                                      cd* __restrict__ d_Re, cd* __restrict__ d_B,
                                                      double* __restrict__ d_sinr, size_t batch) {
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= batch * (size_t)pp.K) return;
    const size_t b = item / pp.K;
    const int k = (int)(item - b * pp.K);
    const int cols = pp.sum_nt + pp.n_ext;
    int r0 = 0;
    for (int i = 0; i < k; ++i) r0 += pp.nr[i];
    const int nrk = pp.nr[k];
    // this receiver's row block, path loss applied (amplitude = sqrt of the power ratio)
    cd Hk[kMuMaxAnt][kMuMaxTx + kMuMaxExt];
    for (int r = 0; r < nrk; ++r)
        for (int c = 0; c < cols; ++c) {
            cd h = d_bigH[(b * pp.sum_nr + r0 + r) * cols + c];
            if (d_pl) {
                const double a = sqrt(d_pl[(size_t)(r0 + r) * cols + c]);
                h.x *= a;
                h.y *= a;
            }
            Hk[r][c] = h;
        }
    // Re_k = pe * ext ext^H + noise_var * I
    cd Re[kMuMaxAnt][kMuMaxAnt];
    for (int r = 0; r < nrk; ++r)
        for (int c = 0; c < nrk; ++c) {
            cd acc = make_double2(0.0, 0.0);
            for (int e = 0; e < pp.n_ext; ++e) acc = mu_macc(acc, Hk[r][pp.sum_nt + e], Hk[c][pp.sum_nt + e]);
            acc.x *= pp.pe;
            acc.y *= pp.pe;
            if (r == c) acc.x += pp.noise_var;
            Re[r][c] = acc;
        }
    // G_j = H F_j for every transmitter, accumulated into the two covariance sums
    cd first[kMuMaxAnt][kMuMaxAnt], Q[kMuMaxAnt][kMuMaxAnt], Gk[kMuMaxAnt][kMuMaxAnt];
    for (int r = 0; r < nrk; ++r)
        for (int c = 0; c < nrk; ++c) first[r][c] = Q[r][c] = make_double2(0.0, 0.0);
    int t0 = 0;
    for (int j = 0; j < pp.K; ++j) {
        const cd* Fj = d_F + ((b * pp.K + j) * kMuMaxTx) * kMuMaxAnt;
        const int rows = pp.joint ? pp.sum_nt : pp.nt[j];
        const int c0 = pp.joint ? 0 : t0;
        cd G[kMuMaxAnt][kMuMaxAnt];
        for (int r = 0; r < nrk; ++r)
            for (int s = 0; s < pp.ns[j]; ++s) {
                cd acc = make_double2(0.0, 0.0);
                for (int t = 0; t < rows; ++t) acc = mu_mac(acc, Hk[r][c0 + t], Fj[t * kMuMaxAnt + s]);
                G[r][s] = acc;
                if (j == k) Gk[r][s] = acc;
            }
        for (int r = 0; r < nrk; ++r)
            for (int c = 0; c < nrk; ++c) {
                cd acc = make_double2(0.0, 0.0);
                for (int s = 0; s < pp.ns[j]; ++s) acc = mu_macc(acc, G[r][s], G[c][s]);
                first[r][c].x += acc.x;
                first[r][c].y += acc.y;
                if (j != k) {
                    Q[r][c].x += acc.x;
                    Q[r][c].y += acc.y;
                }
            }
        t0 += pp.nt[j];
    }
    const size_t o44 = (b * pp.K + k) * kMuMaxAnt * kMuMaxAnt;
    for (int r = 0; r < kMuMaxAnt; ++r)
        for (int c = 0; c < kMuMaxAnt; ++c) {
            const bool in = r < nrk && c < nrk;
            if (d_Q) d_Q[o44 + r * kMuMaxAnt + c] = in ? make_double2(Q[r][c].x + Re[r][c].x, Q[r][c].y + Re[r][c].y)
                                                       : make_double2(0.0, 0.0);
            if (d_Re) d_Re[o44 + r * kMuMaxAnt + c] = in ? Re[r][c] : make_double2(0.0, 0.0);
        }
    for (int l = 0; l < kMuMaxAnt; ++l) {
        const bool live = l < pp.ns[k];
        cd Bl[kMuMaxAnt][kMuMaxAnt];
        for (int r = 0; r < nrk; ++r)
            for (int c = 0; c < nrk; ++c) {
                cd v = make_double2(first[r][c].x + Re[r][c].x, first[r][c].y + Re[r][c].y);
                if (live) {
                    const cd p = mu_macc(make_double2(0.0, 0.0), Gk[r][l], Gk[c][l]);
                    v.x -= p.x;
                    v.y -= p.y;
                }
                Bl[r][c] = v;
            }
        if (d_B)
            for (int r = 0; r < kMuMaxAnt; ++r)
                for (int c = 0; c < kMuMaxAnt; ++c)
                    d_B[((b * pp.K + k) * kMuMaxAnt + l) * kMuMaxAnt * kMuMaxAnt + r * kMuMaxAnt + c] =
                        (live && r < nrk && c < nrk) ? Bl[r][c] : make_double2(0.0, 0.0);
        if (d_sinr && d_U) {
            double out = 0.0;
            if (live) {
                const cd* Uk = d_U + (b * pp.K + k) * kMuMaxAnt * kMuMaxAnt;
                cd aux = make_double2(0.0, 0.0), den = make_double2(0.0, 0.0);
                for (int r = 0; r < nrk; ++r) aux = mu_macc(aux, Gk[r][l], Uk[r * kMuMaxAnt + l]);     // conj(u)^T g, conjugated
                for (int r = 0; r < nrk; ++r) {
                    cd t = make_double2(0.0, 0.0);
                    for (int c = 0; c < nrk; ++c) t = mu_mac(t, Bl[r][c], Uk[c * kMuMaxAnt + l]);
                    den = mu_macc(den, t, Uk[r * kMuMaxAnt + l]);                                      // u^H B u
                }
                const double num = aux.x * aux.x + aux.y * aux.y;
                out = num / sqrt(den.x * den.x + den.y * den.y);                                       // |num / den|
            }
            d_sinr[(b * pp.K + k) * kMuMaxAnt + l] = out;
        }
    }
}

}  // namespace mcle

using namespace mcle;

extern "C" int mcle_mu_link_stats(mcle_ctx* ctx, const mcle_mu_stats_cfg* cfg, const void* d_bigH, const double* d_pathloss,
                                  const void* d_F, const void* d_U, void* d_Q, void* d_Re, void* d_B, double* d_sinr,
                                  size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && cfg != nullptr && d_bigH != nullptr, "null argument");
    MCLE_REQUIRE(cfg->K >= 1 && cfg->K <= kMuMaxK, "K must be in [1, %d]", kMuMaxK);
    MCLE_REQUIRE(cfg->n_ext >= 0 && cfg->n_ext <= kMuMaxExt, "n_ext must be in [0, %d]", kMuMaxExt);
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "Noise variance must be a non-negative value.");
    MuParams pp;
    pp.K = cfg->K;
    pp.n_ext = cfg->n_ext;
    pp.joint = cfg->joint ? 1 : 0;
    pp.noise_var = cfg->noise_var;
    pp.pe = cfg->pe;
    pp.sum_nr = pp.sum_nt = 0;
    for (int k = 0; k < kMuMaxK; ++k) {
        pp.nr[k] = pp.nt[k] = pp.ns[k] = 0;
        if (k >= cfg->K) continue;
        MCLE_REQUIRE(cfg->nr[k] >= 1 && cfg->nr[k] <= kMuMaxAnt && cfg->nt[k] >= 1 && cfg->nt[k] <= kMuMaxAnt,
                     "antenna counts must be in [1, %d]", kMuMaxAnt);
        MCLE_REQUIRE(cfg->ns[k] >= 0 && cfg->ns[k] <= kMuMaxAnt, "stream counts must be in [0, %d]", kMuMaxAnt);
        pp.nr[k] = cfg->nr[k];
        pp.nt[k] = cfg->nt[k];
        pp.ns[k] = cfg->ns[k];
        pp.sum_nr += cfg->nr[k];
        pp.sum_nt += cfg->nt[k];
    }
    bool any_streams = false;
    for (int k = 0; k < cfg->K; ++k) any_streams = any_streams || pp.ns[k] > 0;
    MCLE_REQUIRE(!any_streams || d_F != nullptr, "precoders are needed when any user has streams");
    MCLE_REQUIRE(d_sinr == nullptr || d_U != nullptr, "SINRs need the receive filters");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t items = batch * (size_t)cfg->K;
    hipLaunchKernelGGL(k_mu_link_stats, dim3((unsigned)((items + 63) / 64)), dim3(64), 0, ctx->stream, pp,
                       (const double2*)d_bigH, d_pathloss, (const double2*)d_F, (const double2*)d_U, (double2*)d_Q,
                       (double2*)d_Re, (double2*)d_B, d_sinr, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}
