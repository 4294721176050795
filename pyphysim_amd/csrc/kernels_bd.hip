// kernels_bd.hip -- block diagonalisation of a multi-user MIMO downlink, water-filling, and the fused
// CoMP pipeline built on them.
//
// Reference: comm/waterfilling.py:15-92 (doWF); comm/blockdiagonalization.py:272-363
// (_calc_BD_matrix_no_power_scaling), :365-464 (global / normalised water-filling power scaling), :466-508
// (block_diagonalize), :510-566 (block_diagonalize_no_waterfilling), :568-585 (calc_receive_filter);
// channels/multiuser.py:256-292 (path loss), :1179-1221 (corrupt_concatenated_data);
// apps/comp_BD/simulate_comp_simple.py:95-140 (the application).
//
// The reference takes two SVDs per user (null space of the other users' rows, then the equivalent
// channel).  For the square case (as many transmit as receive antennas -- the only one in which the
// reference's stream count, blockdiagonalization.py:332, equals the null-space dimension) the same
// precoder follows from ONE LQ factorisation H = L Q shared by all users:
//     H^-1 = Q^H M,  M = L^-1 (lower triangular);   Z_k = Q^H M_k  (columns of user k)
//     Z_k^H Z_k = M_k^H M_k = [(H H^H)^-1]_kk = (H_k P0_k H_k^H)^-1      (Schur complement)
// so with the thin SVD M_k = U S V^H (one-sided Jacobi on r <= 4 columns) the right singular vectors of
// the equivalent channel H_k V0_k are Q^H U and its singular values 1 / S.  Singular vectors are
// unique up to a phase per column; the largest entry of every precoder column is made real positive
// (oracle/bd.py: canonical_columns).  newH = H Ms has orthogonal columns inside each user's block, so
// pinv(newH) is its scaled conjugate transpose, with zero rows for streams the water-filling switched off
// (what numpy's pinv returns for a zero column).
#include "modem.hpp"
#include "philox.hpp"
#include "pkcx.hpp"
#include "pipe_common.hpp"
#include "totals.hpp"
#include "wave_draws.hpp"
#include "walk_f64.hpp"
#include "bd_static.hpp"

namespace mcle {

using cd = double2;
constexpr int kBdMaxN = 8;                      // total antennas per side
constexpr int kBdMaxR = 4;                      // antennas per user
constexpr int kWfMaxN = 64;

__device__ __forceinline__ double bd_abs2(cd z) { return z.x * z.x + z.y * z.y; }

// doWF (waterfilling.py:15-92).  Channels sorted by descending gain (ascending stable sort reversed, which is
// what np.argsort(...)[::-1] gives for these sizes); the worst channel is dropped until the powers that
// touch its level fit the budget; the remainder is shared equally.  Sums run left to right like Python's sum.
__device__ __noinline__ void bd_waterfill(const double* gains, int n, double total_power, double nv, double* P,
                                          double* mu) {
    int ord[kWfMaxN];
    for (int i = 0; i < n; ++i) ord[i] = i;
    for (int i = 1; i < n; ++i) {               // insertion sort, ascending, stable
        const int v = ord[i];
        int j = i - 1;
        while (j >= 0 && gains[ord[j]] > gains[v]) {
            ord[j + 1] = ord[j];
            --j;
        }
        ord[j + 1] = v;
    }
    // descending position p <-> ord[n - 1 - p]
    int removed = 0;
    double sum = 0.0, level = 0.0;
    for (;;) {
        const int m = n - removed;
        level = nv / gains[ord[n - m]];         // worst of the remaining channels: descending position m - 1
        sum = 0.0;
        for (int p = 0; p < m; ++p) sum += level - nv / gains[ord[n - 1 - p]];
        if (sum > total_power && removed < n - 1)
            ++removed;
        else
            break;
    }
    const int kept = n - removed;
    const double share = (total_power - sum) / kept;
    for (int i = 0; i < n; ++i) P[i] = 0.0;
    for (int p = 0; p < kept; ++p) P[ord[n - 1 - p]] = share + (level - nv / gains[ord[n - 1 - p]]);
    if (mu) *mu = P[ord[n - 1]] + nv / gains[ord[n - 1]];
}

// One-sided (Hestenes) Jacobi on the columns of A [rows x cols, row-major]: on return the columns are mutually
// orthogonal (A <- A V); V [cols x cols] (may be NULL) accumulates the rotations.
__device__ __noinline__ void bd_jacobi(cd* A, int rows, int cols, cd* V) {
    if (V)
        for (int i = 0; i < cols; ++i)
            for (int c = 0; c < cols; ++c) V[i * cols + c] = mk<double>(i == c ? 1.0 : 0.0, 0.0);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < cols - 1; ++p)
            for (int q = p + 1; q < cols; ++q) {
                double alpha = 0, beta = 0;
                cd gam = mk<double>(0, 0);
                for (int i = 0; i < rows; ++i) {
                    alpha += bd_abs2(A[i * cols + p]);
                    beta += bd_abs2(A[i * cols + q]);
                    gam = cadd(gam, cmulc(A[i * cols + q], A[i * cols + p]));     // a_p^H a_q
                }
                const double g = sqrt(bd_abs2(gam));
                const double rel = g / (sqrt(alpha * beta) + 1e-300);
                off = fmax(off, rel);
                // orthogonal to rounding (or a zero column): leave the pair alone
                if (!(rel >= 1e-15) || !(g > 0.0)) continue;
                const cd ph = mk<double>(gam.x / g, -gam.y / g);                  // e^{-j phi}
                const double zeta = (beta - alpha) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < rows; ++i) {
                    const cd ap = A[i * cols + p], aq = cmul(A[i * cols + q], ph);
                    A[i * cols + p] = csub(cscale(ap, c), cscale(aq, s));
                    A[i * cols + q] = cadd(cscale(ap, s), cscale(aq, c));
                }
                if (V)
                    for (int i = 0; i < cols; ++i) {
                        const cd vp = V[i * cols + p], vq = cmul(V[i * cols + q], ph);
                        V[i * cols + p] = csub(cscale(vp, c), cscale(vq, s));
                        V[i * cols + q] = cadd(cscale(vp, s), cscale(vq, c));
                    }
            }
        if (off < 1e-15) break;
    }
}

// Block diagonalisation of H [n x n row-major], n = K * r.  On return Ms [n x n] is the power-scaled precoder
// and sigma [n] the singular values of the users' equivalent channels (ascending inside each user).
// Q and L are n*n scratch.  Returns false for a numerically singular channel.
__device__ __noinline__ bool bd_solve(const cd* H, int K, int r, double iPu, double nv, int waterfill, cd* Q, cd* L,
                                      cd* Ms, double* sigma) {
    const int n = K * r;
    bool ok = true;
    // ---- LQ by modified Gram-Schmidt on the rows, orthogonalised twice ----
    for (int i = 0; i < n; ++i) {
        double h2 = 0.0;
        for (int c = 0; c < n; ++c) {
            Q[i * n + c] = H[i * n + c];
            h2 += bd_abs2(H[i * n + c]);
        }
        for (int j = 0; j < n; ++j) L[i * n + j] = mk<double>(0, 0);
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < i; ++j) {
                cd d = mk<double>(0, 0);
                for (int c = 0; c < n; ++c) d = cadd(d, cmulc(Q[i * n + c], Q[j * n + c]));   // v . conj(q_j)
                for (int c = 0; c < n; ++c) Q[i * n + c] = csub(Q[i * n + c], cmul(d, Q[j * n + c]));
                L[i * n + j] = cadd(L[i * n + j], d);
            }
        double v2 = 0.0;
        for (int c = 0; c < n; ++c) v2 += bd_abs2(Q[i * n + c]);
        if (!(v2 > 1e-26 * h2) || !(h2 > 0.0)) {
            ok = false;
            v2 = 1.0;
        }
        const double nrm = sqrt(v2), inv = 1.0 / nrm;
        L[i * n + i] = mk<double>(nrm, 0.0);
        for (int c = 0; c < n; ++c) Q[i * n + c] = cscale(Q[i * n + c], inv);
    }
    // ---- M = L^-1 in place, column by column ----
    for (int j = 0; j < n; ++j) {
        L[j * n + j] = mk<double>(1.0 / L[j * n + j].x, 0.0);
        for (int i = j + 1; i < n; ++i) {
            cd acc = cmul(L[i * n + j], L[j * n + j]);                 // m = j term: L[i][j] * M[j][j]
            for (int m = j + 1; m < i; ++m) acc = cadd(acc, cmul(L[i * n + m], L[m * n + j]));
            const double inv = -1.0 / L[i * n + i].x;                  // L[i][i] still the original diagonal
            L[i * n + j] = cscale(acc, inv);
        }
    }
    // ---- per user: thin SVD of M_k by one-sided Jacobi, precoder columns Q^H u ----
    for (int k = 0; k < K; ++k) {
        cd A[kBdMaxN * kBdMaxR];               // rows k*r .. n-1 of M's columns k*r .. k*r+r-1 (the rest is zero)
        const int r0 = k * r, rows = n - r0;
        for (int i = 0; i < rows; ++i)
            for (int c = 0; c < r; ++c) A[i * r + c] = (r0 + i >= r0 + c) ? L[(r0 + i) * n + r0 + c] : mk<double>(0, 0);
        bd_jacobi(A, rows, r, nullptr);
        double S[kBdMaxR];
        int col[kBdMaxR];
        for (int c = 0; c < r; ++c) {
            double n2 = 0;
            for (int i = 0; i < rows; ++i) n2 += bd_abs2(A[i * r + c]);
            S[c] = sqrt(n2);
            col[c] = c;
        }
        for (int i = 0; i < r - 1; ++i)                                         // descending S == ascending sigma
            for (int j = i + 1; j < r; ++j)
                if (S[col[j]] > S[col[i]]) {
                    const int t = col[i];
                    col[i] = col[j];
                    col[j] = t;
                }
        for (int jj = 0; jj < r; ++jj) {
            const int c = col[jj];
            const double inv = 1.0 / S[c];
            sigma[r0 + jj] = inv;
            // v = Q^H u, u = A[:, c] / S[c] living on rows r0 ..
            double best = -1.0;
            cd piv = mk<double>(1.0, 0.0);
            for (int m = 0; m < n; ++m) {
                cd v = mk<double>(0, 0);
                for (int i = 0; i < rows; ++i) v = cadd(v, cmulc(A[i * r + c], Q[(r0 + i) * n + m]));  // u_i conj(Q[i][m])
                v = cscale(v, inv);
                Ms[m * n + r0 + jj] = v;
                const double m2 = bd_abs2(v);
                if (m2 > best * (1.0 + 1e-12)) {
                    best = m2;
                    piv = v;
                }
            }
            const double pm = sqrt(bd_abs2(piv));
            const cd rot = mk<double>(piv.x / pm, -piv.y / pm);
            for (int m = 0; m < n; ++m) Ms[m * n + r0 + jj] = cmul(Ms[m * n + r0 + jj], rot);
        }
    }
    // ---- power scaling ----
    if (waterfill) {
        double gains[kBdMaxN], P[kBdMaxN];
        for (int j = 0; j < n; ++j) gains[j] = sigma[j] * sigma[j];
        bd_waterfill(gains, n, K * iPu, nv, P, nullptr);
        double worst = 0.0;
        for (int k = 0; k < K; ++k) {
            double f2 = 0.0;
            for (int j = k * r; j < (k + 1) * r; ++j) {
                const double a = sqrt(P[j]);
                for (int m = 0; m < n; ++m) {
                    Ms[m * n + j] = cscale(Ms[m * n + j], a);
                    f2 += bd_abs2(Ms[m * n + j]);
                }
            }
            worst = fmax(worst, sqrt(f2));
        }
        const double scale = sqrt(iPu) / worst;
        for (int e = 0; e < n * n; ++e) Ms[e] = cscale(Ms[e], scale);
    } else {
        for (int k = 0; k < K; ++k) {
            double f2 = 0.0;
            for (int j = k * r; j < (k + 1) * r; ++j)
                for (int m = 0; m < n; ++m) f2 += bd_abs2(Ms[m * n + j]);
            const double scale = sqrt(iPu) / sqrt(f2);
            for (int j = k * r; j < (k + 1) * r; ++j)
                for (int m = 0; m < n; ++m) Ms[m * n + j] = cscale(Ms[m * n + j], scale);
        }
    }
    return ok;
}

// W = pinv(H Ms) [n x n, block diagonal]: row s = conj(newH[block rows, s]) / |.|^2, zero for a zero column.
__device__ __noinline__ void bd_receive_filter(const cd* H, const cd* Ms, int K, int r, cd* W) {
    const int n = K * r;
    for (int e = 0; e < n * n; ++e) W[e] = mk<double>(0, 0);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < r; ++j) {
            const int s = k * r + j;
            cd b[kBdMaxR];
            double n2 = 0.0;
            for (int a = 0; a < r; ++a) {
                cd acc = mk<double>(0, 0);
                for (int m = 0; m < n; ++m) acc = cadd(acc, cmul(H[(k * r + a) * n + m], Ms[m * n + s]));
                b[a] = acc;
                n2 += bd_abs2(acc);
            }
            if (n2 > 0.0)
                for (int a = 0; a < r; ++a) W[s * n + k * r + a] = mk<double>(b[a].x / n2, -b[a].y / n2);
        }
}

__global__ __launch_bounds__(64) void k_waterfilling(const double* __restrict__ gains, int n, double total_power,
                                                     double nv, double* __restrict__ P, double* __restrict__ mu,
                                                     size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        double g[kWfMaxN], p[kWfMaxN], level;
        for (int i = 0; i < n; ++i) g[i] = gains[b * n + i];
        bd_waterfill(g, n, total_power, nv, p, &level);
        for (int i = 0; i < n; ++i) P[b * n + i] = p[i];
        if (mu) mu[b] = level;
    }
}

__global__ __launch_bounds__(64) void k_block_diagonalize(const cd* __restrict__ Hin, int K, int r, double iPu, double nv,
                                                          int waterfill, cd* __restrict__ Ms_out,
                                                          cd* __restrict__ newH_out, cd* __restrict__ W_out,
                                                          double* __restrict__ sigma_out,
                                                          uint32_t* __restrict__ skipped, size_t batch) {
    const int n = K * r, nn = n * n;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        cd H[kBdMaxN * kBdMaxN], Q[kBdMaxN * kBdMaxN], L[kBdMaxN * kBdMaxN], Ms[kBdMaxN * kBdMaxN];
        double sigma[kBdMaxN];
        for (int e = 0; e < nn; ++e) H[e] = Hin[b * nn + e];
        const bool ok = bd_solve(H, K, r, iPu, nv, waterfill, Q, L, Ms, sigma);
        if (Ms_out)
            for (int e = 0; e < nn; ++e) Ms_out[b * nn + e] = Ms[e];
        if (newH_out)
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < n; ++c) {
                    cd acc = mk<double>(0, 0);
                    for (int m = 0; m < n; ++m) acc = cadd(acc, cmul(H[i * n + m], Ms[m * n + c]));
                    newH_out[b * nn + i * n + c] = acc;
                }
        if (W_out) {
            bd_receive_filter(H, Ms, K, r, Q);              // Q is free again
            for (int e = 0; e < nn; ++e) W_out[b * nn + e] = Q[e];
        }
        if (sigma_out)
            for (int j = 0; j < n; ++j) sigma_out[b * n + j] = sigma[j];
        if (skipped) skipped[b] = ok ? 0u : 1u;
    }
}

// np.linalg.pinv for small matrices: A [m x n] -> [n x m] = V diag(1/s) U^H over the singular values above
// rcond * max(s) (numpy's default cut-off 1e-15).  With A V = U S from the column Jacobi: pinv = V S^-2 (A V)^H.
__global__ __launch_bounds__(64) void k_pinv(const cd* __restrict__ Ain, int m, int n, double rcond,
                                             cd* __restrict__ out, size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        cd A[kBdMaxN * kBdMaxN], V[kBdMaxN * kBdMaxN];
        double s2[kBdMaxN];
        for (int e = 0; e < m * n; ++e) A[e] = Ain[b * m * n + e];
        bd_jacobi(A, m, n, V);
        double top = 0.0;
        for (int c = 0; c < n; ++c) {
            double v = 0.0;
            for (int i = 0; i < m; ++i) v += bd_abs2(A[i * n + c]);
            s2[c] = v;
            top = fmax(top, v);
        }
        const double cut = rcond * rcond * top;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) {
                cd acc = mk<double>(0, 0);
                for (int c = 0; c < n; ++c)
                    if (s2[c] > cut && s2[c] > 0.0) acc = cadd(acc, cscale(cmulc(V[i * n + c], A[j * n + c]), 1.0 / s2[c]));
                out[b * n * m + i * m + j] = acc;
            }
    }
}

// ---- block diagonalisation with external interference (blockdiagonalization.py:666-1469) --------------------------
// BDWithExtIntBase.calc_whitening_matrices :690-720, WhiteningBD :722-836, EnhancedBD :839-1469;
// channels/multiuser.py:2469-2520 calc_cov_matrix_extint_plus_noise; util/misc.py:1167-1200 calc_whitening_matrix;
// subspace/projections.py:96-130 calcProjectionMatrix.  One lane per channel realization, f64.
struct BdExtParams {
    int K, r, n_ext;        // users, antennas per user (both sides), total antennas of the external interferers
    int method;             // 0: WhiteningBD, 1: EnhancedBD
    int metric;             // EnhancedBD: 0 None, 1 'naive', 2 'fixed', 3 'capacity', 4 report the SINRs of every stream
                            // count (caller-side metrics such as 'effective_throughput'), 5 stream counts given per user
    int num_streams;        // 'naive' / 'fixed'
    int ns_user[4];         // metric 5
    double iPu, nv, pe;
};

// eigen-decomposition of a Hermitian positive semi-definite R [r x r] by one-sided Jacobi on its columns:
// w ascending, eigenvectors in the columns of V (R V = V diag(w)); largest component of each vector real positive
__device__ __noinline__ void bd_heig_psd(const cd* R, int r, double* w, cd* V) {
    cd A[kBdMaxR * kBdMaxR], Vt[kBdMaxR * kBdMaxR];
    for (int e = 0; e < r * r; ++e) A[e] = R[e];
    bd_jacobi(A, r, r, Vt);
    double val[kBdMaxR];
    int ord[kBdMaxR];
    for (int c = 0; c < r; ++c) {
        double n2 = 0.0;
        for (int i = 0; i < r; ++i) n2 += bd_abs2(A[i * r + c]);
        val[c] = sqrt(n2);
        ord[c] = c;
    }
    for (int i = 0; i < r - 1; ++i)
        for (int j = i + 1; j < r; ++j)
            if (val[ord[j]] < val[ord[i]]) {
                const int t = ord[i];
                ord[i] = ord[j];
                ord[j] = t;
            }
    for (int c = 0; c < r; ++c) {
        w[c] = val[ord[c]];
        double best = -1.0;
        cd piv = mk<double>(1.0, 0.0);
        for (int i = 0; i < r; ++i) {
            const cd v = Vt[i * r + ord[c]];
            if (bd_abs2(v) > best * (1.0 + 1e-12)) {
                best = bd_abs2(v);
                piv = v;
            }
        }
        const double pm = sqrt(bd_abs2(piv));
        const cd rot = mk<double>(piv.x / pm, -piv.y / pm);
        for (int i = 0; i < r; ++i) V[i * r + c] = cmul(Vt[i * r + ord[c]], rot);
    }
}

// out [n x m] = pinv(A [m x n]) (numpy's cut-off); the k_pinv arithmetic as a device function
__device__ __noinline__ void bd_pinv_small(const cd* Ain, int m, int n, cd* out) {
    cd A[kBdMaxR * kBdMaxR], V[kBdMaxR * kBdMaxR];
    double s2[kBdMaxR];
    for (int e = 0; e < m * n; ++e) A[e] = Ain[e];
    bd_jacobi(A, m, n, V);
    double top = 0.0;
    for (int c = 0; c < n; ++c) {
        double v = 0.0;
        for (int i = 0; i < m; ++i) v += bd_abs2(A[i * n + c]);
        s2[c] = v;
        top = fmax(top, v);
    }
    const double cut = 1e-30 * top;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            cd acc = mk<double>(0, 0);
            for (int c = 0; c < n; ++c)
                if (s2[c] > cut && s2[c] > 0.0) acc = cadd(acc, cscale(cmulc(V[i * n + c], A[j * n + c]), 1.0 / s2[c]));
            out[i * m + j] = acc;
        }
}

// Outputs per realization b and user k, zero padded: Ms [b][K][n][r] (the user's precoder MsPk: n x Ns),
// W [b][K][r][r] (receive filter, Ns x r), Ns [b][K], cand [b][K][r][r] (metric 4: row ns-1 = the SINRs with ns streams)
__global__ __launch_bounds__(64) void k_bd_extint(BdExtParams pp, const cd* __restrict__ bigH, cd* __restrict__ Ms_out,
                                                  cd* __restrict__ W_out, int32_t* __restrict__ ns_out,
                                                  double* __restrict__ cand_out, uint32_t* __restrict__ skipped,
                                                  size_t batch) {
    const int K = pp.K, r = pp.r, n = K * r, cols = n + pp.n_ext;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        const cd* Hb = bigH + b * (size_t)n * cols;
        cd H[kBdMaxN * kBdMaxN], Q[kBdMaxN * kBdMaxN], L[kBdMaxN * kBdMaxN], Ms[kBdMaxN * kBdMaxN];
        double sigma[kBdMaxN];
        bool ok = true;
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < n; ++c) H[i * n + c] = Hb[(size_t)i * cols + c];
        auto cov = [&](int k, cd* Re) {       // Re_k = pe H_ext,k H_ext,k^H + nv I
            for (int i = 0; i < r; ++i)
                for (int j = 0; j < r; ++j) {
                    cd acc = mk<double>(0, 0);
                    for (int e = 0; e < pp.n_ext; ++e)
                        acc = cadd(acc, cmulc(Hb[(size_t)(k * r + i) * cols + n + e], Hb[(size_t)(k * r + j) * cols + n + e]));
                    acc = cscale(acc, pp.pe);
                    if (i == j) acc.x += pp.nv;
                    Re[i * r + j] = acc;
                }
        };
        auto emit = [&](int k, const cd* MsPk, const cd* W, int ns) {     // MsPk [n x ns], W [ns x r]
            cd* mo = Ms_out + ((b * K + k) * (size_t)n) * r;
            cd* wo = W_out + ((b * K + k) * (size_t)r) * r;
            for (int m = 0; m < n; ++m)
                for (int c = 0; c < r; ++c) mo[m * r + c] = c < ns ? MsPk[m * ns + c] : mk<double>(0, 0);
            for (int i = 0; i < r; ++i)
                for (int c = 0; c < r; ++c) wo[i * r + c] = i < ns ? W[i * r + c] : mk<double>(0, 0);
            if (ns_out) ns_out[b * K + k] = ns;
        };
        if (pp.method == 0) {
            // WhiteningBD: whiten every user's rows with W_k^H = diag(L^-1/2) V^H of eig(Re_k), block-diagonalise the
            // whitened channel, receive filter = pinv(newH) x whitening filter (:781-836)
            cd Wf[4][kBdMaxR * kBdMaxR];
            cd Heq[kBdMaxN * kBdMaxN];
            for (int k = 0; k < K; ++k) {
                cd Re[kBdMaxR * kBdMaxR], V[kBdMaxR * kBdMaxR];
                double w[kBdMaxR];
                cov(k, Re);
                bd_heig_psd(Re, r, w, V);
                for (int i = 0; i < r; ++i) {
                    ok = ok && w[i] > 0.0;
                    const double s = 1.0 / sqrt(w[i] > 0.0 ? w[i] : 1.0);
                    for (int j = 0; j < r; ++j) Wf[k][i * r + j] = cscale(cconj(V[j * r + i]), s);
                }
                for (int i = 0; i < r; ++i)
                    for (int c = 0; c < n; ++c) {
                        cd acc = mk<double>(0, 0);
                        for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Wf[k][i * r + j], H[(k * r + j) * n + c]));
                        Heq[(k * r + i) * n + c] = acc;
                    }
            }
            ok = bd_solve(Heq, K, r, pp.iPu, pp.nv, 0, Q, L, Ms, sigma) && ok;
            bd_receive_filter(Heq, Ms, K, r, Q);                 // pinv(newH), block diagonal
            for (int k = 0; k < K; ++k) {
                cd MsPk[kBdMaxN * kBdMaxR], W[kBdMaxR * kBdMaxR];
                for (int m = 0; m < n; ++m)
                    for (int c = 0; c < r; ++c) MsPk[m * r + c] = Ms[m * n + k * r + c];
                for (int i = 0; i < r; ++i)
                    for (int c = 0; c < r; ++c) {
                        cd acc = mk<double>(0, 0);
                        for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Q[(k * r + i) * n + k * r + j], Wf[k][j * r + c]));
                        W[i * r + c] = acc;
                    }
                emit(k, MsPk, W, r);
            }
        } else if (pp.metric == 0) {
            // EnhancedBD without a metric: plain BD, every user inverts its own block (:1140-1195)
            ok = bd_solve(H, K, r, pp.iPu, pp.nv, 0, Q, L, Ms, sigma);
            bd_receive_filter(H, Ms, K, r, Q);
            for (int k = 0; k < K; ++k) {
                cd MsPk[kBdMaxN * kBdMaxR], W[kBdMaxR * kBdMaxR];
                for (int m = 0; m < n; ++m)
                    for (int c = 0; c < r; ++c) MsPk[m * r + c] = Ms[m * n + k * r + c];
                for (int i = 0; i < r; ++i)
                    for (int c = 0; c < r; ++c) W[i * r + c] = Q[(k * r + i) * n + k * r + c];
                emit(k, MsPk, W, r);
            }
        } else {
            // EnhancedBD with stream reduction (:1197-1411): unit-norm BD directions Ms_bad, then per user a reduction
            // matrix Pk onto the directions least hit by the external interference, power renormalised, receive filter
            // pinv(Pbar Heq_red) Pbar with Pbar the projector onto span(Pk)
            ok = bd_solve(H, K, r, (double)r, pp.nv, 0, Q, L, Ms, sigma);
            for (int k = 0; k < K; ++k) {
                cd Re[kBdMaxR * kBdMaxR], V[kBdMaxR * kBdMaxR], Heq[kBdMaxR * kBdMaxR];
                double w[kBdMaxR];
                cov(k, Re);
                bd_heig_psd(Re, r, w, V);                        // ascending: column 0 = least interfered direction
                for (int i = 0; i < r; ++i)
                    for (int c = 0; c < r; ++c) {
                        cd acc = mk<double>(0, 0);
                        for (int m = 0; m < n; ++m) acc = cadd(acc, cmul(H[(k * r + i) * n + m], Ms[m * n + k * r + c]));
                        Heq[i * r + c] = acc;
                    }
                int lo = 1, hi = r;
                if (pp.metric == 1 || pp.metric == 2) lo = hi = pp.num_streams;
                if (pp.metric == 5) lo = hi = pp.ns_user[k];
                double best_val = -1e300;
                int best_ns = 0;
                cd best_Ms[kBdMaxN * kBdMaxR], best_W[kBdMaxR * kBdMaxR];
                for (int ns = lo; ns <= hi; ++ns) {
                    cd Pk[kBdMaxR * kBdMaxR];                    // r x ns
                    const bool identity = pp.metric == 1 || (pp.metric != 2 && ns == r);
                    for (int i = 0; i < r; ++i)
                        for (int c = 0; c < ns; ++c)
                            Pk[i * ns + c] = identity ? mk<double>(i == c ? 1.0 : 0.0, 0.0) : V[i * r + c];
                    cd MsPk[kBdMaxN * kBdMaxR];
                    double f2 = 0.0;
                    for (int m = 0; m < n; ++m)
                        for (int c = 0; c < ns; ++c) {
                            cd acc = mk<double>(0, 0);
                            for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Ms[m * n + k * r + j], Pk[j * ns + c]));
                            MsPk[m * ns + c] = acc;
                            f2 += bd_abs2(acc);
                        }
                    const double inv_norm = sqrt(pp.iPu) / sqrt(f2);
                    for (int e = 0; e < n * ns; ++e) MsPk[e] = cscale(MsPk[e], inv_norm);
                    cd Hred[kBdMaxR * kBdMaxR], Pbar[kBdMaxR * kBdMaxR], PH[kBdMaxR * kBdMaxR], Pi[kBdMaxR * kBdMaxR];
                    for (int i = 0; i < r; ++i)
                        for (int c = 0; c < ns; ++c) {
                            cd acc = mk<double>(0, 0);
                            for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Heq[i * r + j], Pk[j * ns + c]));
                            Hred[i * ns + c] = cscale(acc, inv_norm);
                        }
                    for (int i = 0; i < r; ++i)                  // Pk has orthonormal columns: Pbar = Pk Pk^H
                        for (int j = 0; j < r; ++j) {
                            cd acc = mk<double>(0, 0);
                            for (int c = 0; c < ns; ++c) acc = cadd(acc, cmulc(Pk[i * ns + c], Pk[j * ns + c]));
                            Pbar[i * r + j] = acc;
                        }
                    for (int i = 0; i < r; ++i)
                        for (int c = 0; c < ns; ++c) {
                            cd acc = mk<double>(0, 0);
                            for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Pbar[i * r + j], Hred[j * ns + c]));
                            PH[i * ns + c] = acc;
                        }
                    bd_pinv_small(PH, r, ns, Pi);                // ns x r
                    cd W[kBdMaxR * kBdMaxR];
                    for (int i = 0; i < ns; ++i)
                        for (int c = 0; c < r; ++c) {
                            cd acc = mk<double>(0, 0);
                            for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(Pi[i * r + j], Pbar[j * r + c]));
                            W[i * r + c] = acc;
                        }
                    double value = 0.0;
                    if (pp.metric == 3 || pp.metric == 4) {      // EnhancedBD._calc_linear_SINRs (:1101-1138)
                        for (int i = 0; i < ns; ++i) {
                            double desired = 0.0, internal = 0.0;
                            for (int c = 0; c < ns; ++c) {
                                cd acc = mk<double>(0, 0);
                                for (int j = 0; j < r; ++j) acc = cadd(acc, cmul(W[i * r + j], Hred[j * ns + c]));
                                if (c == i) desired = bd_abs2(acc);
                                else internal += bd_abs2(acc);
                            }
                            double ext = 0.0;
                            for (int p = 0; p < r; ++p)
                                for (int q = 0; q < r; ++q) ext += cmul(cmul(W[i * r + p], Re[p * r + q]), cconj(W[i * r + q])).x;
                            const double sinr = desired / (internal + fabs(ext));
                            value += log2(1.0 + sinr);
                            if (cand_out && pp.metric == 4) cand_out[((b * K + k) * (size_t)r + (ns - 1)) * r + i] = sinr;
                        }
                    }
                    if (ns == lo || value > best_val) {          // np.argmax: the first maximum
                        best_val = value;
                        best_ns = ns;
                        for (int e = 0; e < n * ns; ++e) best_Ms[e] = MsPk[e];
                        for (int e = 0; e < ns * r; ++e) best_W[e] = W[e];
                    }
                }
                emit(k, best_Ms, best_W, best_ns);
            }
        }
        if (skipped) skipped[b] = ok ? 0u : 1u;
    }
}

struct BdParams {
    int K, r, n_symbols, waterfill, has_pathloss;
    double iPu, noise_var, bd_noise_var;
    double root_pl[16];
};

// Fused CoMP application: one wavefront per chunk of 64 realizations (as k_run_ia).  Phase 1: lane i draws the
// channel of realization i, block-diagonalises it in f64 and parks the receive side of the link in LDS: the
// user's block of W = pinv(H Ms) for each stream (R entries) and d_s = (W H Ms)_ss -- 1 on an active stream
// up to rounding, 0 on a stream the water-filling switched off.  The off-diagonal entries of W H Ms are the
// rounding residue of an exact block diagonalisation (<= 1e-15 |d|) and are not carried.  Phase 2: the wave runs
// the realization's symbol columns: est_s = d_s x_s + W_s . (sigma n_user), demodulate, count.
// R = antennas per user (compile time: every register index below is static).
// Two launches since round 2 (the same split as the IA pipeline, kernels_ia.hip): the per-lane solve keeps four n x n
// complex f64 work matrices in scratch and takes every register (one wave per SIMD), the symbol walk is light and wants
// many resident waves.  The record between them: d[n], W[n][R], flag.
template <typename T, int R>
__global__ __launch_bounds__(64) void k_bd_solve_links(BdParams pp, uint64_t seed, uint64_t first, uint64_t count,
                                                       cx<T>* __restrict__ recs) {
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const int K = pp.K, n = K * R;
    const int stride = n * (R + 1) + 1;
    const Rng rng(seed, first + rl);
    cd H[kBdMaxN * kBdMaxN], Q[kBdMaxN * kBdMaxN], L[kBdMaxN * kBdMaxN], Ms[kBdMaxN * kBdMaxN];
    double sg[kBdMaxN];
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) {
            cd h = cn_sample<double>(rng, STREAM_CHAN, (uint64_t)(i * n + c), 1.0);
            if (pp.has_pathloss) h = cscale(h, pp.root_pl[(i / R) * K + c / R]);
            H[i * n + c] = h;
        }
    const bool ok = bd_solve(H, K, R, pp.iPu, pp.bd_noise_var, pp.waterfill, Q, L, Ms, sg);
    bd_receive_filter(H, Ms, K, R, Q);               // W in Q
    cx<T>* rec = recs + rl * stride;
    for (int s = 0; s < n; ++s) {
        const int r0 = (s / R) * R;
        cd d = mk<double>(0, 0);                     // (W H Ms)_ss
        for (int a = 0; a < R; ++a) {
            cd hm = mk<double>(0, 0);
            for (int m = 0; m < n; ++m) hm = cadd(hm, cmul(H[(r0 + a) * n + m], Ms[m * n + s]));
            d = cadd(d, cmul(Q[s * n + r0 + a], hm));
            rec[n + s * R + a] = mk<T>((T)Q[s * n + r0 + a].x, (T)Q[s * n + r0 + a].y);
        }
        rec[s] = mk<T>((T)d.x, (T)d.y);
    }
    rec[n * (R + 1)] = mk<T>(ok ? (T)1 : (T)0, (T)0);
}

// The same record from the compile-time-sized solve of bd_static.hpp (K R <= 6): every matrix in registers, no scratch.
template <typename T, int K, int R>
__global__ __launch_bounds__(64) void k_bd_solve_links_static(BdParams pp, uint64_t seed, uint64_t first, uint64_t count,
                                                              cx<T>* __restrict__ recs) {
    constexpr int n = K * R, stride = n * (R + 1) + 1;
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const Rng rng(seed, first + rl);
    cd Q[n][n];
#pragma unroll
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < n; ++c) {
            cd h = cn_sample<double>(rng, STREAM_CHAN, (uint64_t)(i * n + c), 1.0);
            if (pp.has_pathloss) h = cscale(h, pp.root_pl[(i / R) * K + c / R]);
            Q[i][c] = h;
        }
    cd d[n], Wb[n][R];
    const bool ok = bd_solve_link_static<K, R>(Q, pp.iPu, pp.bd_noise_var, pp.waterfill, d, Wb);
    cx<T>* rec = recs + rl * stride;
#pragma unroll
    for (int s = 0; s < n; ++s) {
        rec[s] = mk<T>((T)d[s].x, (T)d[s].y);
#pragma unroll
        for (int a = 0; a < R; ++a) rec[n + s * R + a] = mk<T>((T)Wb[s][a].x, (T)Wb[s][a].y);
    }
    rec[n * (R + 1)] = mk<T>(ok ? (T)1 : (T)0, (T)0);
}

// Register budget of the walk: four waves per SIMD (128 VGPRs, 45 of them spilled by the lockstep search) measured against
// three (168 VGPRs, nothing spilled) on the round-3 box: 1.266 vs 1.355-1.365 ms per 131 072 realizations -- the fourth
// wave hides more latency than the spills cost; kept at four.
// Round 4: KC = the number of users as a compile-time constant (0: run-time, arrays and loops sized for kBdMaxN) and MODE = the
// demodulator path chosen by the host (0: demod_one per stream -- slicer, certificate, grid or sweep; 1: the R streams of a
// user swept in lockstep, M <= 8; 2: lockstep through the certificate / candidate grid).  With all three paths and
// eight-entry arrays in one body the complex64 form spilled 45-52 registers at its 128-register bound (profiles/r03:
// VALU busy 0.97 on the draw ledger plus spill traffic).
template <typename T, int R, int KC = 0, int MODE = 0>
// (round 6: workgroups of FOUR independent wavefronts, one flush of the counters -- totals.hpp: wg_flush_waves)
__global__ __launch_bounds__(256, (sizeof(T) == 4 || KC) ? 3 : 2) void k_bd_link(ModemParams<T> mp, BdParams pp, uint64_t seed,
                                                                        uint64_t first, uint64_t count, int per_wave,
                                                                        const cx<T>* __restrict__ recs,
                                                                        mcle_counters* counters,
                                                                        uint32_t* __restrict__ sym_out,
                                                                        uint32_t* __restrict__ bit_out) {
    constexpr int KMAX = KC ? KC : kBdMaxN / R, NMAX = KC ? KC * R : kBdMaxN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = KC ? KC : pp.K, n = K * R;
    const int stride = n * (R + 1) + 1;
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(smem);
    __shared__ cx<T> s_table[256];
    __shared__ float4 s_tab4[sizeof(T) == 4 ? 256 : 1];     // {re, im, |c|^2 / 2, 0}: the lockstep searches of modem.hpp
    __shared__ WgTotals totals_all[4];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WgTotals& totals = totals_all[wv];
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];   // complex128 Box-Muller tables (bm_f64.hpp)
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, (int)blockDim.x);
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    if constexpr (sizeof(T) == 4)
        for (int m = threadIdx.x; m < mp.M; m += blockDim.x) {
            const float2 c = mp.g_table[m];
            s_tab4[m] = make_float4(c.x, c.y, 0.5f * (c.x * c.x + c.y * c.y), 0.f);
        }
    const int lane = threadIdx.x & 63;
    const T sigma = (T)sqrt(pp.noise_var);
    const uint32_t mask = (uint32_t)(mp.M - 1);
    const int NS = pp.n_symbols;
    // f32, min-distance: the R streams of a user searched in lockstep -- directly for constellations of <= 8 points,
    // through the candidate grid otherwise (same decisions as demod_one, one LDS round trip per candidate for all R)
    if (lane == 0) wg_zero(totals);
    __syncthreads();
    const uint64_t n_chunks = (count + per_wave - 1) / per_wave;
    for (uint64_t ch = (uint64_t)blockIdx.x * 4 + wv; ch < n_chunks; ch += (uint64_t)gridDim.x * 4) {
        const uint64_t r_end = (ch + 1) * per_wave < count ? (ch + 1) * per_wave : count;
        for (uint64_t rl = ch * per_wave; rl < r_end; ++rl) {
            const Rng rng(seed, first + rl);
            const cx<T>* D = recs + rl * stride;     // wave-uniform record: scalar loads
            const cx<T>* W = D + n;
            const bool ok = D[n * (R + 1)].x != (T)0;
            unsigned se = 0, be = 0;
            // one symbol column
            auto column = [&](const int (&tx)[NMAX], const cx<T> (&nz)[NMAX]) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (KC || k < K) {
                        cx<T> est[R];
#pragma unroll
                        for (int jj = 0; jj < R; ++jj) {
                            const int s = k * R + jj;
                            if constexpr (sizeof(T) == 4) {          // explicit packed forms (pkcx.hpp): 2 + 2 R instructions
                                pk2 e = pk_cmul(to_pk(D[s]), to_pk(s_table[tx[s]]));
#pragma unroll
                                for (int a = 0; a < R; ++a) e = pk_cfma(to_pk(W[s * R + a]), to_pk(nz[k * R + a]), e);
                                est[jj] = from_pk(e);
                            } else {
                                est[jj] = cmul(D[s], s_table[tx[s]]);
#pragma unroll
                                for (int a = 0; a < R; ++a) est[jj] = cfma4(W[s * R + a], nz[k * R + a], est[jj]);
                            }
                        }
                        int dec[R];
                        if constexpr (sizeof(T) == 4 && MODE == 1) {
                            demod_mindist_multi<R>(s_tab4, mp.M, est, dec);
                        } else if constexpr (sizeof(T) == 4 && MODE == 2) {
                            demod_multi_cert(mp, est, dec, [&](int (&d_)[R]) { demod_grid4_multi<R>(s_tab4, s_grid, mp.grid, mp.M, est, d_); });
                        } else {
#pragma unroll
                            for (int jj = 0; jj < R; ++jj) dec[jj] = demod_one(mp, s_table, s_grid, est[jj]);
                        }
#pragma unroll
                        for (int jj = 0; jj < R; ++jj) {
                            const unsigned x = (unsigned)(tx[k * R + jj] ^ dec[jj]);
                            se += (x != 0u);
                            be += __popc(x);
                        }
                    }
            };
            if ((NS & 1) == 0) {
                // two columns per lane and pass: whole Philox blocks, symbol blocks shared across the wave
                for (int t0 = 0; t0 < NS; t0 += kPairCols) {
                    const int t = t0 + 2 * lane;
                    int ta[NMAX], tb[NMAX];
                    wave_symbol_pairs<NMAX>(rng, n, (uint32_t)NS, (uint32_t)t0, mask, lane, ta, tb);
                    if (t < NS) {
                        if constexpr (sizeof(T) == 8) {
                            // complex128: user by user -- the noise of ONE user's R antennas (both columns), its estimates, its
                            // decisions -- instead of all K R antennas' draws first: 8 R live noise registers instead of 8 K R,
                            // so that the registers can be bounded for three wavefronts per SIMD (round 5; k_ia_link alike)
#pragma unroll
                            for (int k = 0; k < KMAX; ++k)
                                if (KC || k < K) {
                                    cx<T> za[R], zb[R];
#pragma unroll
                                    for (int a = 0; a < R; ++a)
                                        cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)(k * R + a) * (uint32_t)NS + (uint32_t)t) >> 1, sigma,
                                                    za[a], zb[a], s_bm);
#pragma unroll
                                    for (int jj = 0; jj < R; ++jj) {
                                        const int s = k * R + jj;
                                        cx<T> ea = cmul(D[s], s_table[ta[s]]), eb = cmul(D[s], s_table[tb[s]]);
#pragma unroll
                                        for (int a = 0; a < R; ++a) {
                                            ea = cfma4(W[s * R + a], za[a], ea);
                                            eb = cfma4(W[s * R + a], zb[a], eb);
                                        }
                                        const unsigned da = (unsigned)(ta[s] ^ demod_one(mp, s_table, s_grid, ea));
                                        const unsigned db = (unsigned)(tb[s] ^ demod_one(mp, s_table, s_grid, eb));
                                        se += (da != 0u) + (db != 0u);
                                        be += __popc(da) + __popc(db);
                                    }
                                }
                        } else {
                            cx<T> za[NMAX], zb[NMAX];
#pragma unroll
                            for (int a = 0; a < NMAX; ++a)
                                if (KC || a < n)
                                    cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)a * (uint32_t)NS + (uint32_t)t) >> 1, sigma,
                                                za[a], zb[a], s_bm);
                            column(ta, za);
                            column(tb, zb);
                        }
                    }
                }
            } else {
                for (int t = lane; t < NS; t += 64) {
                    int tx[NMAX];
                    cx<T> nz[NMAX];
#pragma unroll
                    for (int a = 0; a < NMAX; ++a)
                        if (KC || a < n) {
                            tx[a] = (int)symbol_at(rng, (uint64_t)a * NS + t, mask);   // randint(0, M, [n, NSymbs])
                            nz[a] = cn_sample<T>(rng, STREAM_NOISE, (uint64_t)a * NS + t, sigma);
                        }
                    column(tx, nz);
                }
            }
            se = wave_sum_u32(se);
            be = wave_sum_u32(be);
            if (lane == 0) wg_account(totals, se, be, !ok, rl, sym_out, bit_out);
        }
    }
    wg_flush_waves<4>(totals_all, counters, (unsigned long long)n * NS, (unsigned long long)n * NS * mp.bits);
}

template <typename T, int R>
static int launch_run_bd(mcle_ctx* ctx, const mcle_bd_cfg* cfg, const BdParams& pp, uint64_t seed, uint64_t first,
                         uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    const int n = cfg->K * R;
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    int rc;
    void* recs = nullptr;
    const uint64_t kSlice = 1ull << 20;          // realizations per solve + walk pair: bounds the record buffer
    const uint64_t slice = count < kSlice ? count : kSlice;
    const size_t stride = (size_t)(n * (R + 1) + 1);
    if ((rc = ctx->scratch((size_t)slice * stride * sizeof(cx<T>), &recs))) return rc;
    const size_t lds = (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long);
    const int per_wave = 8;
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t m = count - off < slice ? count - off : slice;
        const dim3 sgrid((unsigned)((m + 63) / 64));
        bool launched = false;
#define MCLE_BD_STATIC(K_)                                                                                                \
    if (!launched && cfg->K == K_ && K_ * R <= 6) {                                                                       \
        hipLaunchKernelGGL((k_bd_solve_links_static<T, K_, (K_ * R <= 6 ? R : 1)>), sgrid, dim3(64), 0, ctx->stream, pp, seed,  \
                           first + off, m, (cx<T>*)recs);                                                                \
        launched = true;                                                                                                  \
    }
        if (!ctx->opt[MCLE_OPT_BD_RUNTIME_SOLVE]) {       // compile-time-sized solve (registers) where n = K R <= 6
            MCLE_BD_STATIC(2) MCLE_BD_STATIC(3) MCLE_BD_STATIC(4) MCLE_BD_STATIC(5) MCLE_BD_STATIC(6)
        }
#undef MCLE_BD_STATIC
        if (!launched)
            hipLaunchKernelGGL((k_bd_solve_links<T, R>), sgrid, dim3(64), 0, ctx->stream, pp, seed, first + off, m,
                               (cx<T>*)recs);
        MCLE_LAUNCH_CHECK();
        // (complex128: three wavefronts per SIMD where the user count is a compile-time constant -- 168 registers, nothing spilled --
        //  two for the run-time-sized form, which spills 18 - 58 registers at that bound)
        const bool kc = R <= 2 && (cfg->K == 2 || cfg->K == 3) && cfg->K * R <= kBdMaxN;
        // the resident set: the wavefronts per SIMD of k_bd_link's __launch_bounds__ (3 for complex64 and the compile-time user
        // counts, 2 otherwise) -- rounds 4-5 sized the complex64 grid for four and ran a partial second wave of workgroups
        const uint64_t cap = (uint64_t)ctx->n_cu * ((sizeof(T) == 4 || kc) ? 3 : 2);       // workgroups of four wavefronts
        const uint64_t chunks = (m + per_wave - 1) / per_wave;
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, cap, (chunks + 3) / 4, 2);     // a chunk is 8-16 realizations; one chunk per workgroup measured 1.5 x slower
        // demodulator path of the walk (compile-time in the kernel): complex64 min-distance in lockstep -- a sweep for M <= 8,
        // certificate / candidate grid otherwise; everything else one demod_one per stream
        // (round 5: a constellation WITH a certificate -- square QAM, QPSK -- goes through it whatever its size: the application's
        //  own 4-PSK was swept point by point in lockstep, ~20 instructions per decision where the quadrant certificate takes 6)
        const int mode = (sizeof(T) == 4 && mp.method == MCLE_DEMOD_MINDIST)
                             ? ((mp.cert && mp.grid.G > 0) ? 2 : (mp.M <= 8 ? 1 : (mp.grid.G > 0 ? 2 : 0))) : 0;
        bool walked = false;
#define MCLE_BD_WALK(KC_, MODE_)                                                                                          \
    if (!walked && (KC_ == 0 || (cfg->K == KC_ && KC_ * R <= kBdMaxN)) && mode == MODE_) {                                  \
        hipLaunchKernelGGL((k_bd_link<T, R, (KC_ * R <= kBdMaxN ? KC_ : 0), (sizeof(T) == 4 ? MODE_ : 0)>), dim3(grid), dim3(256), lds, \
                           ctx->stream, mp, pp, seed, first + off, m, per_wave, (const cx<T>*)recs, d_counters,           \
                           d_sym_err ? d_sym_err + off : nullptr, d_bit_err ? d_bit_err + off : nullptr);                 \
        walked = true;                                                                                                    \
    }
        if constexpr (R <= 2) {
            // two or three users, an even number of columns >= 128: the packed walk of walk_f64.hpp (round 6; complex64 since its last day)
            if (kc && link_walk_f64_fits(cfg->n_symbols) && !ctx->opt[MCLE_OPT_WALK_LEGACY]) {
#define MCLE_BD_PACKED(KC_, ABL_)                                                                                             \
    if (!walked && cfg->K == KC_) {                                                                                           \
        launch_link_walk<T, BdWalk<KC_, R>, ABL_>(ctx, mp, cfg->n_symbols, pp.noise_var, seed, first + off, m, (const cx<T>*)recs, \
                                                   d_counters, d_sym_err ? d_sym_err + off : nullptr,                        \
                                                   d_bit_err ? d_bit_err + off : nullptr);                                   \
        walked = true;                                                                                                        \
    }
#ifdef MCLE_EXPERIMENTS
                if constexpr (R == 2 && sizeof(T) == 8) {
                    switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) {
                        case 1: MCLE_BD_PACKED(3, 1) break;
                        case 2: MCLE_BD_PACKED(3, 2) break;
                        case 4: MCLE_BD_PACKED(3, 4) break;
                        case 6: MCLE_BD_PACKED(3, 6) break;
                        case 8: MCLE_BD_PACKED(3, 8) break;
                        case 16: MCLE_BD_PACKED(3, 16) break;
                        case 31: MCLE_BD_PACKED(3, 31) break;
                        default: break;
                    }
                }
#endif
                MCLE_BD_PACKED(2, 0) MCLE_BD_PACKED(3, 0)
#undef MCLE_BD_PACKED
            }
        }
        if constexpr (R <= 2) {                    // users as a compile-time count where the arrays then shrink: K = 2, 3
            MCLE_BD_WALK(2, 0) MCLE_BD_WALK(2, 1) MCLE_BD_WALK(2, 2) MCLE_BD_WALK(3, 0) MCLE_BD_WALK(3, 1) MCLE_BD_WALK(3, 2)
        }
        MCLE_BD_WALK(0, 0) MCLE_BD_WALK(0, 1) MCLE_BD_WALK(0, 2)
#undef MCLE_BD_WALK
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

static int check_bd_dims(int K, int r) {
    MCLE_REQUIRE(K >= 1 && r >= 1 && r <= kBdMaxR && K * r <= kBdMaxN,
                 "block diagonalisation supports num_users * antennas <= %d with at most %d antennas per user "
                 "(got %d users x %d)", kBdMaxN, kBdMaxR, K, r);
    return MCLE_OK;
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_waterfilling(mcle_ctx* ctx, const double* d_gains, int n, double total_power, double noise_var,
                      double* d_powers, double* d_mu, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && (batch == 0 || (d_gains != nullptr && d_powers != nullptr)), "null argument");
    MCLE_REQUIRE(n >= 1 && n <= kWfMaxN, "number of parallel channels must be in [1, %d] (got %d)", kWfMaxN, n);
    MCLE_REQUIRE(total_power > 0.0 && noise_var >= 0.0, "total power must be positive and the noise variance "
                                                        "non-negative");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    hipLaunchKernelGGL(k_waterfilling, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream, d_gains, n,
                       total_power, noise_var, d_powers, d_mu, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_block_diagonalize(mcle_ctx* ctx, const void* d_H, int num_users, int n_rx_per_user, double iPu,
                           double noise_var, int waterfilling, void* d_Ms, void* d_newH, void* d_W, double* d_sigma,
                           uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && (batch == 0 || d_H != nullptr), "null argument");
    int rc = check_bd_dims(num_users, n_rx_per_user);
    if (rc) return rc;
    MCLE_REQUIRE(iPu > 0.0, "the power per user must be positive");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    if (batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    hipLaunchKernelGGL(k_block_diagonalize, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream,
                       (const double2*)d_H, num_users, n_rx_per_user, iPu, noise_var, waterfilling ? 1 : 0,
                       (double2*)d_Ms, (double2*)d_newH, (double2*)d_W, d_sigma, d_skipped, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_pinv(mcle_ctx* ctx, const void* d_A, int m, int n, double rcond, void* d_out, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && (batch == 0 || (d_A != nullptr && d_out != nullptr)), "null argument");
    MCLE_REQUIRE(m >= 1 && n >= 1 && m <= kBdMaxN && n <= kBdMaxN, "pinv supports matrices up to %d x %d (got %d x %d)",
                 kBdMaxN, kBdMaxN, m, n);
    MCLE_REQUIRE(rcond >= 0.0, "rcond must be non-negative");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    hipLaunchKernelGGL(k_pinv, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream, (const double2*)d_A, m, n,
                       rcond, (double2*)d_out, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_bd_extint(mcle_ctx* ctx, const mcle_bd_extint_cfg* cfg, const void* d_bigH, void* d_Ms, void* d_W,
                   int32_t* d_ns, double* d_cand_sinr, uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && cfg != nullptr && d_bigH != nullptr && d_Ms != nullptr && d_W != nullptr, "null argument");
    MCLE_REQUIRE(cfg->num_users >= 1 && cfg->num_users <= 4 && cfg->n_ant_per_user >= 1 && cfg->n_ant_per_user <= kBdMaxR &&
                     cfg->num_users * cfg->n_ant_per_user <= kBdMaxN,
                 "block diagonalisation covers up to 4 users x 4 antennas with at most %d antennas in total", kBdMaxN);
    MCLE_REQUIRE(cfg->n_ext >= 0 && cfg->n_ext <= 8, "at most 8 external-interference antennas");
    MCLE_REQUIRE(cfg->method == 0 || cfg->method == 1, "method: 0 = WhiteningBD, 1 = EnhancedBD");
    MCLE_REQUIRE(cfg->metric >= 0 && cfg->metric <= 5, "metric must be in [0, 5]");
    MCLE_REQUIRE(cfg->iPu > 0.0 && cfg->noise_var >= 0.0 && cfg->pe >= 0.0, "iPu must be positive, noise_var and pe non-negative");
    MCLE_REQUIRE(cfg->method == 1 || cfg->noise_var > 0.0 || cfg->n_ext >= cfg->n_ant_per_user,
                 "whitening needs a positive definite interference-plus-noise covariance");
    if (cfg->method == 1 && (cfg->metric == 1 || cfg->metric == 2))
        MCLE_REQUIRE(cfg->num_streams >= 1 && cfg->num_streams <= cfg->n_ant_per_user,
                     "num_streams must be in [1, %d]", cfg->n_ant_per_user);
    if (cfg->method == 1 && cfg->metric == 5)
        for (int k = 0; k < cfg->num_users; ++k)
            MCLE_REQUIRE(cfg->ns_user[k] >= 1 && cfg->ns_user[k] <= cfg->n_ant_per_user, "ns_user[%d] out of range", k);
    MCLE_REQUIRE(cfg->metric != 4 || d_cand_sinr != nullptr, "metric 4 reports the candidate SINRs: d_cand_sinr is required");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    BdExtParams pp;
    pp.K = cfg->num_users;
    pp.r = cfg->n_ant_per_user;
    pp.n_ext = cfg->n_ext;
    pp.method = cfg->method;
    pp.metric = cfg->metric;
    pp.num_streams = cfg->num_streams;
    for (int k = 0; k < 4; ++k) pp.ns_user[k] = cfg->ns_user[k];
    pp.iPu = cfg->iPu;
    pp.nv = cfg->noise_var;
    pp.pe = cfg->pe;
    if (d_cand_sinr)
        MCLE_HIP(hipMemsetAsync(d_cand_sinr, 0, batch * (size_t)pp.K * pp.r * pp.r * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_bd_extint, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream, pp, (const cd*)d_bigH,
                       (cd*)d_Ms, (cd*)d_W, d_ns, d_cand_sinr, d_skipped, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_run_bd(mcle_ctx* ctx, int dtype, const mcle_bd_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    if ((rc = check_bd_dims(cfg->K, cfg->nr))) return rc;
    MCLE_REQUIRE(cfg->n_symbols >= 1, "n_symbols must be positive");
    MCLE_REQUIRE(cfg->iPu > 0.0, "the power per user must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0 && cfg->bd_noise_var >= 0.0, "noise variances must be non-negative");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    BdParams pp;
    pp.K = cfg->K;
    pp.r = cfg->nr;
    pp.n_symbols = cfg->n_symbols;
    pp.waterfill = cfg->waterfilling ? 1 : 0;
    pp.has_pathloss = cfg->has_pathloss ? 1 : 0;
    pp.iPu = cfg->iPu;
    pp.noise_var = cfg->noise_var;
    pp.bd_noise_var = cfg->bd_noise_var;
    for (int i = 0; i < 16; ++i) {
        const bool used = pp.has_pathloss && i < cfg->K * cfg->K;      // [K][K] row-major: rx user, tx user
        if (used) MCLE_REQUIRE(cfg->pathloss[i] >= 0.0, "path loss must be non-negative");
        pp.root_pl[i] = used ? std::sqrt(cfg->pathloss[i]) : 1.0;
    }
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    switch (cfg->nr * 2 + (dtype == MCLE_F64)) {
        case 2: return launch_run_bd<float, 1>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 3: return launch_run_bd<double, 1>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 4: return launch_run_bd<float, 2>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 5: return launch_run_bd<double, 2>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 6: return launch_run_bd<float, 3>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 7: return launch_run_bd<double, 3>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        case 8: return launch_run_bd<float, 4>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
        default: return launch_run_bd<double, 4>(ctx, cfg, pp, seed, first, count, d_counters, d_sym_err, d_bit_err);
    }
}

}  // extern "C"
