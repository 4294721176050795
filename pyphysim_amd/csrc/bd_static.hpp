// bd_static.hpp -- the block diagonalisation of kernels_bd.hip (bd_solve + bd_receive_filter, reference
// comm/blockdiagonalization.py:272-566, comm/waterfilling.py:15-92) with compile-time sizes, for the per-lane solve of the
// fused CoMP pipeline (k_bd_solve_links).
//
// The run-time-sized form keeps four n x n complex f64 work matrices in private arrays with dynamic indices: 5 KB of
// scratch per lane, and the solve kernel was bound by that traffic (round 2: 0.77 ms of a 2.05 ms step, 37 %).  With K and R
// template parameters every loop unrolls and every index is a constant, so the matrices are registers:
//   * the channel is orthogonalised IN PLACE (H -> Q); H itself is not needed afterwards because
//     H Ms_s = L Q Q^H u_s = L u_s: the user's block of H Ms is the R x R diagonal block of L (kept aside before L is
//     inverted) times the top R entries of the singular vector u_s;
//   * L is stored as its lower triangle only (the unrolled code never touches the rest).
// The precoder Ms itself is never formed: the link needs only its column norms (power scaling) and H Ms.
// n = K R <= 6 keeps it inside the registers of a one-wave-per-SIMD launch (Q 144 + L 84 + blocks 36 + utop 48 ...);
// larger geometries stay on the run-time-sized path.  Same operation order as bd_solve wherever the two overlap.
#pragma once
#include "common.hpp"

namespace mcle {

using cd_s = double2;
__device__ __forceinline__ double bds_abs2(cd_s z) { return z.x * z.x + z.y * z.y; }

// doWF with static indices: descending-gain order by rank counting (ties keep index order reversed, like
// np.argsort(...)[::-1] on a stable ascending sort), then the same removal loop as bd_waterfill.
template <int N>
__device__ __forceinline__ void bd_waterfill_static(const double (&gains)[N], double total_power, double nv, double (&P)[N]) {
    // ascending stable position of i: #j with g[j] < g[i], or g[j] == g[i] and j < i
    double asc[N];          // gains in ascending stable order
    int pos[N];             // position of channel i in that order
#pragma unroll
    for (int i = 0; i < N; ++i) {
        int p = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) p += (gains[j] < gains[i] || (gains[j] == gains[i] && j < i)) ? 1 : 0;
        pos[i] = p;
    }
#pragma unroll
    for (int p = 0; p < N; ++p) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) v = pos[i] == p ? gains[i] : v;
        asc[p] = v;
    }
    // descending position q <-> asc[N - 1 - q]
    int removed = 0;
    double sum = 0.0, level = 0.0;
#pragma unroll
    for (int it = 0; it < N; ++it) {           // at most N - 1 removals; once settled the state no longer changes
        const int m = N - removed;
        double worst = 0.0;
#pragma unroll
        for (int p = 0; p < N; ++p) worst = (p == N - m) ? asc[p] : worst;
        const double lv = nv / worst;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q)
            if (q < m) s += lv - nv / asc[N - 1 - q];
        level = lv;
        sum = s;
        if (s > total_power && removed < N - 1) ++removed;
        else break;
    }
    const int kept = N - removed;
    const double share = (total_power - sum) / kept;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int q = N - 1 - pos[i];          // descending position of channel i
        P[i] = q < kept ? share + (level - nv / gains[i]) : 0.0;
    }
}

// One-sided Jacobi on the COLS columns of A [ROWS x COLS] (bd_jacobi without the accumulated rotations)
template <int ROWS, int COLS>
__device__ __forceinline__ void bd_jacobi_static(cd_s (&A)[ROWS][COLS]) {
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < COLS - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < COLS; ++q) {
                double alpha = 0, beta = 0;
                cd_s gam = mk<double>(0, 0);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) {
                    alpha += bds_abs2(A[i][p]);
                    beta += bds_abs2(A[i][q]);
                    gam = cadd(gam, cmulc(A[i][q], A[i][p]));
                }
                const double g = sqrt(bds_abs2(gam));
                const double rel = g / (sqrt(alpha * beta) + 1e-300);
                off = fmax(off, rel);
                if (!(rel >= 1e-15) || !(g > 0.0)) continue;
                const cd_s ph = mk<double>(gam.x / g, -gam.y / g);
                const double zeta = (beta - alpha) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < ROWS; ++i) {
                    const cd_s ap = A[i][p], aq = cmul(A[i][q], ph);
                    A[i][p] = csub(cscale(ap, c), cscale(aq, s));
                    A[i][q] = cadd(cscale(ap, s), cscale(aq, c));
                }
            }
        if (off < 1e-15) break;
    }
}

// One user's share of bd_solve: thin SVD of M's columns r0 .. r0 + R - 1 (rows r0 .. N - 1), precoder columns Q^H u made
// canonical (largest entry real positive), sigma ascending inside the user; utop = the first R entries of every u (scaled
// and rotated like its precoder column), for the user's block of H Ms = L_kk utop.
template <int K, int R, int KU>
__device__ __forceinline__ void bd_user_static(const cd_s (&Q)[K * R][K * R], const cd_s (&L)[K * R][K * R],
                                               double (&cn2)[K * R], double (&sigma)[K * R], cd_s (&utop)[K * R][R]) {
    constexpr int N = K * R, r0 = KU * R, ROWS = N - r0;
    cd_s A[ROWS][R];
#pragma unroll
    for (int i = 0; i < ROWS; ++i)
#pragma unroll
        for (int c = 0; c < R; ++c) A[i][c] = (i >= c) ? L[r0 + i][r0 + c] : mk<double>(0, 0);
    bd_jacobi_static<ROWS, R>(A);
    double S[R];
    int rank[R];                               // descending-S position of column c (ties: lower column first)
#pragma unroll
    for (int c = 0; c < R; ++c) {
        double n2 = 0;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) n2 += bds_abs2(A[i][c]);
        S[c] = sqrt(n2);
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        int p = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) p += (S[j] > S[c] || (S[j] == S[c] && j < c)) ? 1 : 0;
        rank[c] = p;
    }
#pragma unroll
    for (int jj = 0; jj < R; ++jj) {
        // the column with rank jj, selected without a dynamic index
        cd_s u[ROWS];
        double Sc = S[0];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) u[i] = A[i][0];
#pragma unroll
        for (int c = 1; c < R; ++c)
            if (rank[c] == jj) {
                Sc = S[c];
#pragma unroll
                for (int i = 0; i < ROWS; ++i) u[i] = A[i][c];
            }
        const double inv = 1.0 / Sc;
        sigma[r0 + jj] = inv;
        double best = -1.0;
        cd_s piv = mk<double>(1.0, 0.0);
        cd_s col[N];
#pragma unroll
        for (int m = 0; m < N; ++m) {
            cd_s v = mk<double>(0, 0);
#pragma unroll
            for (int i = 0; i < ROWS; ++i) v = cadd(v, cmulc(u[i], Q[r0 + i][m]));
            v = cscale(v, inv);
            col[m] = v;
            const double m2 = bds_abs2(v);
            if (m2 > best * (1.0 + 1e-12)) {
                best = m2;
                piv = v;
            }
        }
        const double pm = sqrt(bds_abs2(piv));
        const cd_s rot = mk<double>(piv.x / pm, -piv.y / pm);
        double c2 = 0.0;                       // squared norm of the precoder column (1 up to rounding: Q is unitary)
#pragma unroll
        for (int m = 0; m < N; ++m) c2 += bds_abs2(cmul(col[m], rot));
        cn2[r0 + jj] = c2;
#pragma unroll
        for (int i = 0; i < R; ++i) utop[r0 + jj][i] = cmul(cscale(u[i], inv), rot);
    }
}

template <int K, int R, int KU>
struct BdUsers {
    __device__ static __forceinline__ void run(const cd_s (&Q)[K * R][K * R], const cd_s (&L)[K * R][K * R],
                                               double (&cn2)[K * R], double (&sigma)[K * R], cd_s (&utop)[K * R][R]) {
        BdUsers<K, R, KU - 1>::run(Q, L, cn2, sigma, utop);
        bd_user_static<K, R, KU>(Q, L, cn2, sigma, utop);
    }
};
template <int K, int R>
struct BdUsers<K, R, -1> {
    __device__ static __forceinline__ void run(const cd_s (&)[K * R][K * R], const cd_s (&)[K * R][K * R],
                                               double (&)[K * R], double (&)[K * R], cd_s (&)[K * R][R]) {}
};

// Q: in = the channel H [N x N], out = the orthonormal rows of its LQ factorisation.  Out:
// d[s] = (W H Ms)_ss and Wb[s][a] = the user's block of row s of W = pinv(H Ms).  Returns false for a singular channel.
template <int K, int R>
__device__ __forceinline__ bool bd_solve_link_static(cd_s (&Q)[K * R][K * R], double iPu, double nv, int waterfill,
                                                     cd_s (&d)[K * R], cd_s (&Wb)[K * R][R]) {
    constexpr int N = K * R;
    cd_s L[N][N];                 // lower triangle used
    cd_s Ld[N][R];                // Ld[r0 + a][i] = L[r0 + a][r0 + i], i <= a: the diagonal blocks before the inversion
    double cn2[N];                // squared norms of the precoder columns (the precoder itself is not needed for the link)
    cd_s utop[N][R];
    double sigma[N];
    bool ok = true;
    // ---- LQ by modified Gram-Schmidt on the rows, orthogonalised twice, in place ----
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double h2 = 0.0;
#pragma unroll
        for (int c = 0; c < N; ++c) h2 += bds_abs2(Q[i][c]);
#pragma unroll
        for (int j = 0; j < i; ++j) L[i][j] = mk<double>(0, 0);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int j = 0; j < i; ++j) {
                cd_s dd = mk<double>(0, 0);
#pragma unroll
                for (int c = 0; c < N; ++c) dd = cadd(dd, cmulc(Q[i][c], Q[j][c]));
#pragma unroll
                for (int c = 0; c < N; ++c) Q[i][c] = csub(Q[i][c], cmul(dd, Q[j][c]));
                L[i][j] = cadd(L[i][j], dd);
            }
        double v2 = 0.0;
#pragma unroll
        for (int c = 0; c < N; ++c) v2 += bds_abs2(Q[i][c]);
        if (!(v2 > 1e-26 * h2) || !(h2 > 0.0)) {
            ok = false;
            v2 = 1.0;
        }
        const double nrm = sqrt(v2), inv = 1.0 / nrm;
        L[i][i] = mk<double>(nrm, 0.0);
#pragma unroll
        for (int c = 0; c < N; ++c) Q[i][c] = cscale(Q[i][c], inv);
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int c = 0; c < R; ++c) Ld[i][c] = ((i / R) * R + c <= i) ? L[i][(i / R) * R + c] : mk<double>(0, 0);
    // ---- M = L^-1 in place, column by column ----
#pragma unroll
    for (int j = 0; j < N; ++j) {
        L[j][j] = mk<double>(1.0 / L[j][j].x, 0.0);
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            cd_s acc = cmul(L[i][j], L[j][j]);
#pragma unroll
            for (int m = j + 1; m < i; ++m) acc = cadd(acc, cmul(L[i][m], L[m][j]));
            const double inv = -1.0 / L[i][i].x;         // L[i][i] is still the original diagonal (column i comes later)
            L[i][j] = cscale(acc, inv);
        }
    }
    // ---- per user: thin SVD of M_k, precoder columns ----
    BdUsers<K, R, K - 1>::run(Q, L, cn2, sigma, utop);
    // ---- power scaling: fac[s] = the real factor column s of Ms (and of utop) is multiplied by ----
    double fac[N];
    if (waterfill) {
        double gains[N], P[N];
#pragma unroll
        for (int j = 0; j < N; ++j) gains[j] = sigma[j] * sigma[j];
        bd_waterfill_static<N>(gains, K * iPu, nv, P);
        double worst = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double f2 = 0.0;
#pragma unroll
            for (int j = k * R; j < (k + 1) * R; ++j) {
                const double a = sqrt(P[j]);
                fac[j] = a;
                f2 += a * a * cn2[j];
            }
            worst = fmax(worst, sqrt(f2));
        }
        const double scale = sqrt(iPu) / worst;
#pragma unroll
        for (int j = 0; j < N; ++j) fac[j] *= scale;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double f2 = 0.0;
#pragma unroll
            for (int j = k * R; j < (k + 1) * R; ++j) f2 += cn2[j];
            const double scale = sqrt(iPu) / sqrt(f2);
#pragma unroll
            for (int j = k * R; j < (k + 1) * R; ++j) fac[j] = scale;
        }
    }
    // ---- receive side: b = the user's block of H Ms column s = L_kk utop_s fac_s; W row = conj(b) / |b|^2 ----
#pragma unroll
    for (int s = 0; s < N; ++s) {
        const int r0 = (s / R) * R;
        cd_s b[R];
        double n2 = 0.0;
#pragma unroll
        for (int a = 0; a < R; ++a) {
            cd_s acc = mk<double>(0, 0);
#pragma unroll
            for (int i = 0; i <= a; ++i) acc = cadd(acc, cmul(Ld[r0 + a][i], utop[s][i]));
            acc = cscale(acc, fac[s]);
            b[a] = acc;
            n2 += bds_abs2(acc);
        }
        cd_s dd = mk<double>(0, 0);
#pragma unroll
        for (int a = 0; a < R; ++a) {
            const cd_s w = n2 > 0.0 ? mk<double>(b[a].x / n2, -b[a].y / n2) : mk<double>(0, 0);
            Wb[s][a] = w;
            dd = cadd(dd, cmul(w, b[a]));
        }
        d[s] = dd;
    }
    return ok;
}

}  // namespace mcle
