// mimo_svd.hpp -- SVD / GMD precoders and receive filters as register-resident device functions, shared by the
// operator kernels (kernels_mimo.hip) and the fused flat-fading MIMO pipeline (pipeline_mimo_flat.hip).
#pragma once
#include "mimo.hpp"

namespace mcle {

// ---- SVD by one-sided (Hestenes) Jacobi in f64: H = U diag(S) V^H, S descending ---------------------
// On return A[:, c] = U[:, c] * S[c].  Singular vectors carry the usual per-pair phase freedom (a
// consistent pair, not LAPACK's particular choice).
template <int NA>
__device__ __forceinline__ void jacobi_svd(double2 (&A)[NA][NA], double2 (&V)[NA][NA], double (&S)[NA]) {
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) V[r][c] = mk<double>(r == c ? 1.0 : 0.0, 0.0);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < NA - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < NA; ++q) {
                double alpha = 0, beta = 0;
                double2 gam = mk<double>(0, 0);
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    alpha += A[r][p].x * A[r][p].x + A[r][p].y * A[r][p].y;
                    beta += A[r][q].x * A[r][q].x + A[r][q].y * A[r][q].y;
                    gam = cadd(gam, cmulc(A[r][q], A[r][p]));  // a_p^H a_q
                }
                const double g = sqrt(gam.x * gam.x + gam.y * gam.y);
                const double rel = g / (sqrt(alpha * beta) + 1e-300);
                off = fmax(off, rel);
                // a pair that is orthogonal to rounding is left alone: its inner product is noise, and rotating
                // by the PHASE of noise would turn the column by an arbitrary angle
                if (rel < 1e-15) continue;
                const double2 ph = mk<double>(gam.x / g, gam.y / g);  // e^{j phi}
                const double zeta = (beta - alpha) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const double2 ap = A[r][p], aq = cmul(A[r][q], cconj(ph));
                    A[r][p] = csub(cscale(ap, c), cscale(aq, s));
                    A[r][q] = cadd(cscale(ap, s), cscale(aq, c));
                    const double2 vp = V[r][p], vq = cmul(V[r][q], cconj(ph));
                    V[r][p] = csub(cscale(vp, c), cscale(vq, s));
                    V[r][q] = cadd(cscale(vp, s), cscale(vq, c));
                }
            }
        if (off < 1e-15) break;
    }
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        double n2 = 0;
#pragma unroll
        for (int r = 0; r < NA; ++r) n2 += A[r][c].x * A[r][c].x + A[r][c].y * A[r][c].y;
        S[c] = sqrt(n2);
    }
    // canonical phases: the largest-magnitude entry of every right singular vector real and positive (first on
    // ties), the left vector turned with it -- the pair (U, V) then depends continuously on H, so two builds of
    // this routine (operator kernel, fused pipeline) agree to rounding
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        double best = -1.0;
        double2 piv = mk<double>(1.0, 0.0);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const double m2 = V[r][c].x * V[r][c].x + V[r][c].y * V[r][c].y;
            if (m2 > best * (1.0 + 1e-12)) {
                best = m2;
                piv = V[r][c];
            }
        }
        const double m = sqrt(piv.x * piv.x + piv.y * piv.y);
        const double2 rot = mk<double>(piv.x / m, -piv.y / m);   // conj(piv) / |piv|
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            V[r][c] = cmul(V[r][c], rot);
            A[r][c] = cmul(A[r][c], rot);
        }
    }
    // selection sort of the columns, descending S (predicated swaps keep everything in registers)
#pragma unroll
    for (int i = 0; i < NA - 1; ++i)
#pragma unroll
        for (int j = i + 1; j < NA; ++j)
            if (S[j] > S[i]) {
                const double ts = S[i];
                S[i] = S[j];
                S[j] = ts;
#pragma unroll
                for (int r = 0; r < NA; ++r) {
                    const double2 ta = A[r][i], tv = V[r][i];
                    A[r][i] = A[r][j];
                    A[r][j] = ta;
                    V[r][i] = V[r][j];
                    V[r][j] = tv;
                }
            }
}

// SVDMimo (mimo.py:833-946): W = V / sqrt(Nt), G = diag(1/S) U^H sqrt(Nt) for square H (consumed in place).
template <int NA>
__device__ __forceinline__ void svd_filters_dev(double2 (&A)[NA][NA], double2 (&W)[NA][NA], double2 (&G)[NA][NA],
                                                double (&S)[NA]) {
    double2 V[NA][NA];
    jacobi_svd<NA>(A, V, S);
    const double root = sqrt((double)NA);
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        const double k = root / (S[c] * S[c]);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            // W[:, c] = V[:, c] / sqrt(Nt);  G[c, :] = conj(U[:, c]) sqrt(Nt) / S = conj(A[:, c]) sqrt(Nt) / S^2
            W[r][c] = mk<double>(V[r][c].x / root, V[r][c].y / root);
            G[c][r] = mk<double>(A[r][c].x * k, -A[r][c].y * k);
        }
    }
}

// GMDMimo (mimo.py:952-1067) with util.misc.gmd (misc.py:18-159): H = Q R P^H with R upper triangular and
// a constant diagonal (the geometric mean of the singular values).  W = P / sqrt(Nt); the receive filter
// is Blast's (ZF / MMSE) on the equivalent channel Q R.  R depends on S only and is reproduced exactly;
// Q and P inherit the SVD's phase freedom.
template <int NA>
__device__ __forceinline__ bool gmd_filters_dev(const double2 (&Hin)[NA][NA], double nv, double2 (&W)[NA][NA],
                                                double2 (&G)[NA][NA], double (&R)[NA][NA]) {
    double2 Q[NA][NA], P[NA][NA];
    double d[NA], z[NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            Q[r][c] = Hin[r][c];
            R[r][c] = 0.0;
        }
    jacobi_svd<NA>(Q, P, d);
#pragma unroll
    for (int c = 0; c < NA; ++c) {  // Q <- U
        z[c] = 0.0;
#pragma unroll
        for (int r = 0; r < NA; ++r) Q[r][c] = cscale(Q[r][c], 1.0 / d[c]);
    }
    double prod = 1.0;
#pragma unroll
    for (int c = 0; c < NA; ++c) prod *= d[c];
    const double sigma_bar = pow(prod, 1.0 / NA);
    int perm[NA], invperm[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) perm[c] = invperm[c] = c;
    int large = 1, small = NA - 1;
#pragma unroll
    for (int k = 0; k < NA - 1; ++k) {
        bool flag = false;
        int i;
        // (runtime indexes into the small arrays are resolved with predicated scans)
        auto at = [&](const double(&arr)[NA], int idx) {
            double v = 0;
#pragma unroll
            for (int c = 0; c < NA; ++c)
                if (c == idx) v = arr[c];
            return v;
        };
        auto iat = [&](const int(&arr)[NA], int idx) {
            int v = 0;
#pragma unroll
            for (int c = 0; c < NA; ++c)
                if (c == idx) v = arr[c];
            return v;
        };
        if (d[k] >= sigma_bar) {
            i = iat(perm, small);
            --small;
            if (at(d, i) >= sigma_bar) flag = true;
        } else {
            i = iat(perm, large);
            ++large;
            if (at(d, i) <= sigma_bar) flag = true;
        }
        const int k1 = k + 1;
        if (i != k1) {
            const double di = at(d, i), t = d[k1];
            d[k1] = di;
            const int j = invperm[k1];
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                if (c == i) d[c] = t;
                if (c == j) perm[c] = i;
                if (c == i) invperm[c] = j;
            }
#pragma unroll
            for (int c = 0; c < NA; ++c)
                if (c == i) {
#pragma unroll
                    for (int r = 0; r < NA; ++r) {
                        const double2 tq = Q[r][k1], tp = P[r][k1];
                        Q[r][k1] = Q[r][c];
                        Q[r][c] = tq;
                        P[r][k1] = P[r][c];
                        P[r][c] = tp;
                    }
                }
        }
        const double delta1 = d[k], delta2 = d[k1];
        const double sq1 = delta1 * delta1, sq2 = delta2 * delta2;
        double c = 1.0, s = 0.0;
        if (!flag) {
            c = sqrt((sigma_bar * sigma_bar - sq2) / (sq1 - sq2));
            s = sqrt(1.0 - c * c);
        }
        d[k1] = delta1 * delta2 / sigma_bar;
        z[k] = s * c * (sq2 - sq1) / sigma_bar;
        R[k][k] = sigma_bar;
#pragma unroll
        for (int r = 0; r < NA; ++r)
            if (r < k) {
                R[r][k] = z[r] * c;
                z[r] = -z[r] * s;
            }
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const double2 p0 = P[r][k], p1 = P[r][k1];   // P[:, (k,k1)] @ [[c, -s], [s, c]]
            P[r][k] = cadd(cscale(p0, c), cscale(p1, s));
            P[r][k1] = cadd(cscale(p0, -s), cscale(p1, c));
            const double2 q0 = Q[r][k], q1 = Q[r][k1];   // Q[:, (k,k1)] @ [[c d1, -s d2], [s d2, c d1]] / sigma_bar
            Q[r][k] = cscale(cadd(cscale(q0, c * delta1), cscale(q1, s * delta2)), 1.0 / sigma_bar);
            Q[r][k1] = cscale(cadd(cscale(q0, -s * delta2), cscale(q1, c * delta1)), 1.0 / sigma_bar);
        }
    }
    R[NA - 1][NA - 1] = sigma_bar;
#pragma unroll
    for (int r = 0; r < NA - 1; ++r) R[r][NA - 1] = z[r];
    // equivalent channel Q R and Blast's filter on it (mimo.py:1001-1011)
    double2 Heq[NA][NA];
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            double2 acc = mk<double>(0, 0);
#pragma unroll
            for (int m = 0; m < NA; ++m) acc = cadd(acc, cscale(Q[r][m], R[m][c]));
            Heq[r][c] = acc;
        }
    const bool ok = blast_filter<NA, NA>(Heq, nv, G);
    const double root = sqrt((double)NA);
#pragma unroll
    for (int r = 0; r < NA; ++r)
#pragma unroll
        for (int c = 0; c < NA; ++c) W[r][c] = mk<double>(P[r][c].x / root, P[r][c].y / root);
    return ok;
}

}  // namespace mcle
