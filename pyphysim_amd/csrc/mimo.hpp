// mimo.hpp -- Blast receive filter as a register-resident device function.
// Reference: mimo/mimo.py:264-309 (_calcZeroForceFilter = pinv, _calcMMSEFilter = solve(H^H H +
// sigma^2 I, H^H)) and :597-607 (x sqrt(Nt); MMSE iff noise_var > 0).  For full column rank
// pinv(H) = solve(H^H H, H^H), so one Cholesky solve serves both; a non-positive pivot (rank
// deficiency) is reported so the caller can count the realization as skipped.
// Always f64: ~NT^3 flops per realization, noise next to the FFTs, and it keeps the filter of an
// ill-conditioned channel (cond(H^H H) ~ 1e4 at 25 dB) out of f32 trouble.
#pragma once
#include "common.hpp"
#include "pkcx.hpp"

namespace mcle {

// H: NR x NT (row-major), G: NT x NR.  Returns false if H^H H + nv I is not positive definite.
// R = double everywhere one filter serves a whole realization; R = float only where a filter is needed per
// subcarrier (frequency-selective pipeline, f32 instantiation).
template <typename R, int NT, int NR>
__device__ __forceinline__ bool blast_filter_t(const cx<R> (&H)[NR][NT], R nv, cx<R> (&G)[NT][NR]) {
    typedef cx<R> C;
    C L[NT][NT];  // lower Cholesky factor of A = H^H H + nv I (strict upper part unused)
    R invd[NT];   // 1 / L[j][j]
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = j; i < NT; ++i) {
            C a = mk<R>(0, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) a = cadd(a, cmulc(H[r][j], H[r][i]));  // conj(H[r][i]) * H[r][j]
            // a = A[i][j]
            if (i == j) a.x += nv;
#pragma unroll
            for (int k = 0; k < j; ++k) a = csub(a, cmulc(L[i][k], L[j][k]));
            if (i == j) {
                ok = ok && (a.x > (R)(sizeof(R) == 8 ? 1e-300 : 1e-30));
                L[j][j] = mk<R>(sqrt(a.x), (R)0);
                invd[j] = (R)1 / L[j][j].x;
            } else {
                L[i][j] = cscale(a, invd[j]);
            }
        }
    }
    const R root_nt = (R)sqrt((double)NT);
#pragma unroll
    for (int c = 0; c < NR; ++c) {
        C z[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {  // L z = H^H[:, c]
            C v = cconj(H[c][i]);
#pragma unroll
            for (int k = 0; k < i; ++k) v = csub(v, cmul(L[i][k], z[k]));
            z[i] = cscale(v, invd[i]);
        }
#pragma unroll
        for (int i = NT - 1; i >= 0; --i) {  // L^H w = z
            C v = z[i];
#pragma unroll
            for (int k = i + 1; k < NT; ++k) v = csub(v, cmul(cconj(L[k][i]), z[k]));
            z[i] = cscale(v, invd[i]);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) G[i][c] = cscale(z[i], root_nt);
    }
    return ok;
}

// x = sqrt(NT) (H^H H + nv I)^-1 H^H y without forming the filter: one right-hand side instead of NR.  For
// callers that need a filter per column (frequency-selective channels) -- same decisions as G y.
// f32 instantiation: every complex multiply-accumulate as four chained FMAs (the product-then-add form of the f64 parity
// code is six operations) and one v_rcp / v_rsq per pivot; same algorithm, rounding-level differences only.
template <typename C> __device__ __forceinline__ C cx_macc(C acc, C a, C b) {    // acc + a conj(b)
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.y, b.x, acc.y);
    acc.y = fma(-a.x, b.y, acc.y);
    return acc;
}
template <typename C> __device__ __forceinline__ C cx_msub(C acc, C a, C b) {    // acc - a b
    acc.x = fma(-a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(-a.x, b.y, acc.y);
    acc.y = fma(-a.y, b.x, acc.y);
    return acc;
}
template <typename C> __device__ __forceinline__ C cx_msubc(C acc, C a, C b) {   // acc - a conj(b)
    acc.x = fma(-a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(-a.y, b.x, acc.y);
    acc.y = fma(a.x, b.y, acc.y);
    return acc;
}
template <int NT, int NR>
__device__ __forceinline__ bool blast_solve_f32(const float2 (&H)[NR][NT], float nv, const float2 (&y)[NR], float2 (&x)[NT]) {
    float2 L[NT][NT];
    float invd[NT];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = j; i < NT; ++i) {
            float2 a = make_float2(i == j ? nv : 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < NR; ++r) a = cx_macc(a, H[r][j], H[r][i]);      // conj(H[r][i]) * H[r][j]
#pragma unroll
            for (int k = 0; k < j; ++k) a = cx_msubc(a, L[i][k], L[j][k]);
            if (i == j) {
                ok = ok && (a.x > 1e-30f);
                invd[j] = __builtin_amdgcn_rsqf(a.x);
                L[j][j] = make_float2(a.x * invd[j], 0.f);
            } else {
                L[i][j] = make_float2(a.x * invd[j], a.y * invd[j]);
            }
        }
    }
    float2 z[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {  // L z = H^H y
        float2 v = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NR; ++r) v = cx_macc(v, y[r], H[r][i]);               // conj(H[r][i]) * y[r]
#pragma unroll
        for (int k = 0; k < i; ++k) v = cx_msub(v, L[i][k], z[k]);
        z[i] = make_float2(v.x * invd[i], v.y * invd[i]);
    }
    const float root_nt = (float)sqrt((double)NT);
#pragma unroll
    for (int i = NT - 1; i >= 0; --i) {  // L^H w = z
        float2 v = z[i];
#pragma unroll
        for (int k = i + 1; k < NT; ++k) v = cx_msub(v, make_float2(L[k][i].x, -L[k][i].y), z[k]);
        z[i] = make_float2(v.x * invd[i], v.y * invd[i]);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) x[i] = make_float2(z[i].x * root_nt, z[i].y * root_nt);
    return ok;
}

template <typename R, int NT, int NR>
__device__ __forceinline__ bool blast_solve_t(const cx<R> (&H)[NR][NT], R nv, const cx<R> (&y)[NR], cx<R> (&x)[NT]) {
    if constexpr (sizeof(R) == 4) return blast_solve_f32<NT, NR>(H, nv, y, x);
    typedef cx<R> C;
    C L[NT][NT];
    R invd[NT];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = j; i < NT; ++i) {
            C a = mk<R>(0, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) a = cadd(a, cmulc(H[r][j], H[r][i]));
            if (i == j) a.x += nv;
#pragma unroll
            for (int k = 0; k < j; ++k) a = csub(a, cmulc(L[i][k], L[j][k]));
            if (i == j) {
                ok = ok && (a.x > (R)(sizeof(R) == 8 ? 1e-300 : 1e-30));
                L[j][j] = mk<R>(sqrt(a.x), (R)0);
                invd[j] = (R)1 / L[j][j].x;
            } else {
                L[i][j] = cscale(a, invd[j]);
            }
        }
    }
    C z[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {  // L z = H^H y
        C v = mk<R>(0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) v = cadd(v, cmulc(y[r], H[r][i]));   // conj(H[r][i]) * y[r]
#pragma unroll
        for (int k = 0; k < i; ++k) v = csub(v, cmul(L[i][k], z[k]));
        z[i] = cscale(v, invd[i]);
    }
    const R root_nt = (R)sqrt((double)NT);
#pragma unroll
    for (int i = NT - 1; i >= 0; --i) {  // L^H w = z
        C v = z[i];
#pragma unroll
        for (int k = i + 1; k < NT; ++k) v = csub(v, cmul(cconj(L[k][i]), z[k]));
        z[i] = cscale(v, invd[i]);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) x[i] = cscale(z[i], root_nt);
    return ok;
}

// The same solve from the Gram matrix: A[i][k] (i >= k: lower triangle, real diagonal in .x) = sum_r conj(H[r][i]) H[r][k] and
// b[i] = sum_r conj(H[r][i]) y[r] accumulated ROW BY ROW by the caller (blast_gram_row) -- a kernel that forms H(f) one receive
// antenna at a time (mimo_tdl_wave.hpp) never holds the whole Nr x Nt matrix of a subcarrier, let alone of several.  Same
// algorithm in the same order as blast_solve_t (sum over r ascending, then + nv on the diagonal, Cholesky, two substitutions), every
// complex multiply-add as four chained FMAs in both arithmetics (blast_solve_t<double> keeps the product-then-add form).
template <typename R, int NT>
__device__ __forceinline__ void blast_gram_row(const cx<R> (&h)[NT], cx<R> yr, cx<R> (&A)[NT][NT], cx<R> (&b)[NT]) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        A[k][k].x = fma(h[k].x, h[k].x, fma(h[k].y, h[k].y, A[k][k].x));
#pragma unroll
        for (int i = k + 1; i < NT; ++i) {
            if constexpr (sizeof(R) == 4) A[i][k] = from_pk(pk_cfma_conj(to_pk(h[k]), to_pk(h[i]), to_pk(A[i][k])));
            else A[i][k] = cx_macc(A[i][k], h[k], h[i]);                      // + conj(h[i]) * h[k], four chained FMAs
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if constexpr (sizeof(R) == 4) b[i] = from_pk(pk_cfma_conj(to_pk(yr), to_pk(h[i]), to_pk(b[i])));
        else b[i] = cx_macc(b[i], yr, h[i]);                                  // + conj(h[i]) * y
    }
}
template <typename R, int NT>
__device__ __forceinline__ bool blast_solve_gram(const cx<R> (&A)[NT][NT], R nv, const cx<R> (&b)[NT], cx<R> (&x)[NT]) {
    typedef cx<R> C;
    C L[NT][NT];
    R invd[NT];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = j; i < NT; ++i) {
            C a = A[i][j];
            if (i == j) {
                a.x += nv;
                a.y = 0;
            }
#pragma unroll
            for (int k = 0; k < j; ++k) {
                a = cx_msubc(a, L[i][k], L[j][k]);
            }
            if (i == j) {
                ok = ok && (a.x > (R)(sizeof(R) == 8 ? 1e-300 : 1e-30));
                if constexpr (sizeof(R) == 4) {
                    invd[j] = __builtin_amdgcn_rsqf(a.x);
                    L[j][j] = mk<R>(a.x * invd[j], (R)0);
                } else {                                                    // sqrt and 1 / sqrt by Newton steps (common.hpp)
                    double root, inv;
                    sqrt_rsqrt_newton((double)a.x, root, inv);        // (a failed pivot yields NaNs the caller discards with `ok`)
                    L[j][j] = mk<R>((R)root, (R)0);
                    invd[j] = (R)inv;
                }
            } else {
                L[i][j] = cscale(a, invd[j]);
            }
        }
    }
    C z[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {  // L z = H^H y
        C v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) {
            v = cx_msub(v, L[i][k], z[k]);
        }
        z[i] = cscale(v, invd[i]);
    }
    const R root_nt = (R)sqrt((double)NT);
#pragma unroll
    for (int i = NT - 1; i >= 0; --i) {  // L^H w = z
        C v = z[i];
#pragma unroll
        for (int k = i + 1; k < NT; ++k) {
            v = cx_msub(v, mk<R>(L[k][i].x, -L[k][i].y), z[k]);
        }
        z[i] = cscale(v, invd[i]);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) x[i] = cscale(z[i], root_nt);
    return ok;
}

template <int NT, int NR>
__device__ __forceinline__ bool blast_filter(const double2 (&H)[NR][NT], double nv, double2 (&G)[NT][NR]) {
    return blast_filter_t<double, NT, NR>(H, nv, G);
}

}  // namespace mcle
