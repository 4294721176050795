// mimo.hpp -- Blast receive filter as a register-resident device function.
// Reference: mimo/mimo.py:264-309 (_calcZeroForceFilter = pinv, _calcMMSEFilter = solve(H^H H +
// sigma^2 I, H^H)) and :597-607 (x sqrt(Nt); MMSE iff noise_var > 0).  For full column rank
// pinv(H) = solve(H^H H, H^H), so one Cholesky solve serves both; a non-positive pivot (rank
// deficiency) is reported so the caller can count the realization as skipped.
// Always f64: ~NT^3 flops per realization, noise next to the FFTs, and it keeps the filter of an
// ill-conditioned channel (cond(H^H H) ~ 1e4 at 25 dB) out of f32 trouble.
#pragma once
#include "common.hpp"

namespace mcle {

// H: NR x NT (row-major), G: NT x NR.  Returns false if H^H H + nv I is not positive definite.
template <int NT, int NR>
__device__ __forceinline__ bool blast_filter(const double2 (&H)[NR][NT], double nv, double2 (&G)[NT][NR]) {
    double2 L[NT][NT];  // lower Cholesky factor of A = H^H H + nv I (strict upper part unused)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = j; i < NT; ++i) {
            double2 a = mk<double>(0, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) a = cadd(a, cmulc(H[r][j], H[r][i]));  // conj(H[r][i]) * H[r][j]
            // a = A[i][j]
            if (i == j) a.x += nv;
#pragma unroll
            for (int k = 0; k < j; ++k) a = csub(a, cmulc(L[i][k], L[j][k]));
            if (i == j) {
                ok = ok && (a.x > 1e-300);
                L[j][j] = mk<double>(sqrt(a.x), 0.0);
            } else {
                L[i][j] = cscale(a, 1.0 / L[j][j].x);
            }
        }
    }
    const double root_nt = sqrt((double)NT);
#pragma unroll
    for (int c = 0; c < NR; ++c) {
        double2 z[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {  // L z = H^H[:, c]
            double2 v = cconj(H[c][i]);
#pragma unroll
            for (int k = 0; k < i; ++k) v = csub(v, cmul(L[i][k], z[k]));
            z[i] = cscale(v, 1.0 / L[i][i].x);
        }
#pragma unroll
        for (int i = NT - 1; i >= 0; --i) {  // L^H w = z
            double2 v = z[i];
#pragma unroll
            for (int k = i + 1; k < NT; ++k) v = csub(v, cmul(cconj(L[k][i]), z[k]));
            z[i] = cscale(v, 1.0 / L[i][i].x);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) G[i][c] = cscale(z[i], root_nt);
    }
    return ok;
}

}  // namespace mcle
