// kernels_ia.hip -- closed-form interference alignment for the 3-user 2x2 MIMO interference channel
// (one stream per user) and the fused config-5 pipeline.
//
// Reference: ia/algorithms.py:73-96 (_calc_E), :98-191 (_updateF / _updateW), :194-265 (solve with
// use_best_init), ia/iabase.py:188-200,299-327 (full_F / full_W_H), :768-789,897-996 (SINR),
// channels/multiuser.py:1003-1044,1179-1262 (randomize / corrupt_data), apps/ia/simulate_ia.py:94-245.
//
// Everything is 2x2 and solved in closed form in f64 registers (the solver is ~10^3 flops per
// realization).  np.linalg.eig's eigenvectors come from LAPACK zgeev, which scales each vector to
// unit 2-norm with its largest-magnitude component real and positive; the 2x2 closed form below
// applies the same normalisation, so F_0 -- whose phase rotates the post-filter noise -- matches.
#include <cstdlib>
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"
#include "qam_pack.hpp"
#include "wave_draws.hpp"
#include "walk_f64.hpp"

namespace mcle {

using cd = double2;

struct M2 {
    cd a, b, c, d;  // [[a, b], [c, d]]
};
struct V2 {
    cd x, y;
};

__device__ __forceinline__ cd cdiv_(cd p, cd q) { return cdivide(p, q); }
__device__ __forceinline__ double cabs2(cd z) { return z.x * z.x + z.y * z.y; }
__device__ __forceinline__ cd csqrt_(cd z) {
    const double r = sqrt(sqrt(cabs2(z)));
    const double th = 0.5 * atan2(z.y, z.x);
    double s, c;
    sincos(th, &s, &c);
    return mk<double>(r * c, r * s);
}
__device__ __forceinline__ M2 mmul(const M2& p, const M2& q) {
    M2 r;
    r.a = cadd(cmul(p.a, q.a), cmul(p.b, q.c));
    r.b = cadd(cmul(p.a, q.b), cmul(p.b, q.d));
    r.c = cadd(cmul(p.c, q.a), cmul(p.d, q.c));
    r.d = cadd(cmul(p.c, q.b), cmul(p.d, q.d));
    return r;
}
__device__ __forceinline__ V2 mvec(const M2& p, const V2& v) {
    V2 r;
    r.x = cadd(cmul(p.a, v.x), cmul(p.b, v.y));
    r.y = cadd(cmul(p.c, v.x), cmul(p.d, v.y));
    return r;
}
__device__ __forceinline__ M2 minv(const M2& p, bool& ok) {
    const cd det = csub(cmul(p.a, p.d), cmul(p.b, p.c));
    ok = ok && (cabs2(det) > 1e-280);
    M2 r;
    r.a = cdiv_(p.d, det);
    r.b = cdiv_(mk<double>(-p.b.x, -p.b.y), det);
    r.c = cdiv_(mk<double>(-p.c.x, -p.c.y), det);
    r.d = cdiv_(p.a, det);
    return r;
}
__device__ __forceinline__ V2 vnormalize(V2 v) {
    const double n = 1.0 / sqrt(cabs2(v.x) + cabs2(v.y));
    v.x = cscale(v.x, n);
    v.y = cscale(v.y, n);
    return v;
}
// LAPACK zgeev convention: unit norm, largest-magnitude component real positive (first max on ties)
__device__ __forceinline__ V2 lapack_normalize(V2 v) {
    v = vnormalize(v);
    const cd piv = cabs2(v.y) > cabs2(v.x) ? v.y : v.x;
    const double m = sqrt(cabs2(piv));
    const cd ph = mk<double>(piv.x / m, -piv.y / m);  // conj(piv)/|piv|
    v.x = cmul(v.x, ph);
    v.y = cmul(v.y, ph);
    return v;
}
// eigenvector of E for eigenvalue lam: the better conditioned of the two rows of (E - lam I)
__device__ __forceinline__ V2 eigvec2(const M2& E, cd lam) {
    const V2 v1{E.b, csub(lam, E.a)};
    const V2 v2{csub(lam, E.d), E.c};
    const double n1 = cabs2(v1.x) + cabs2(v1.y), n2 = cabs2(v2.x) + cabs2(v2.y);
    return lapack_normalize(n1 >= n2 ? v1 : v2);
}

struct IaSolution {
    V2 F[3];    // precoders (column vectors, unit norm)
    V2 U[3];    // receive filters full_W_H (row vectors)
    double sinr[3];
    double capacity;
    bool ok;
};

// H[k][l]: channel from transmitter l to receiver k
__device__ __forceinline__ void ia_candidate(const M2 (&H)[3][3], const M2& invH32, const M2& invH23, V2 F0,
                                             double nv, IaSolution& s) {
    s.ok = true;
    s.F[0] = vnormalize(F0);
    s.F[1] = vnormalize(mvec(invH32, mvec(H[2][0], F0)));
    s.F[2] = vnormalize(mvec(invH23, mvec(H[1][0], F0)));
    const V2 a[3] = {mvec(H[0][1], s.F[1]), mvec(H[1][0], s.F[0]), mvec(H[2][0], s.F[0])};
    s.capacity = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // W_k spans the null space of a a^H:  W^H = [a_y, -a_x] / |a|
        const double n = 1.0 / sqrt(cabs2(a[k].x) + cabs2(a[k].y));
        const V2 wh{cscale(a[k].y, n), cscale(mk<double>(-a[k].x.x, -a[k].x.y), n)};
        const V2 hf = mvec(H[k][k], s.F[k]);
        const cd eq = cadd(cmul(wh.x, hf.x), cmul(wh.y, hf.y));  // W^H H_kk F_k
        s.U[k].x = cdiv_(wh.x, eq);
        s.U[k].y = cdiv_(wh.y, eq);
        double den = nv * (cabs2(s.U[k].x) + cabs2(s.U[k].y));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j == k) continue;
            const V2 g = mvec(H[k][j], s.F[j]);
            den += cabs2(cadd(cmul(s.U[k].x, g.x), cmul(s.U[k].y, g.y)));
        }
        s.sinr[k] = 1.0 / den;
        s.capacity += log2(1.0 + s.sinr[k]);
    }
}

// ia_closed_form: not inlined, one compiled body serves the operator kernel and the iterative solver's 'closed_form'
// start.  The pipeline's solve kernel inlines the same body (ia_closed_form_inl): behind a call, H and the solution
// live in scratch (1.3 KB per lane), inlined they stay in registers.
__device__ __forceinline__ IaSolution ia_closed_form_inl(const M2 (&H)[3][3], double nv) {
    bool ok = true;
    const M2 i31 = minv(H[2][0], ok), i12 = minv(H[0][1], ok), i23 = minv(H[1][2], ok), i32 = minv(H[2][1], ok);
    // E = H31^-1 H32 . (H12^-1 H13 . (H23^-1 H21))
    const M2 E = mmul(mmul(i31, H[2][1]), mmul(mmul(i12, H[0][2]), mmul(i23, H[1][0])));
    const cd tr = cadd(E.a, E.d);
    const cd det = csub(cmul(E.a, E.d), cmul(E.b, E.c));
    const cd disc = csqrt_(csub(cmul(tr, tr), cscale(det, 4.0)));
    const cd l0 = cscale(cadd(tr, disc), 0.5), l1 = cscale(csub(tr, disc), 0.5);
    IaSolution s0, s1;
    ia_candidate(H, i32, i23, eigvec2(E, l0), nv, s0);
    ia_candidate(H, i32, i23, eigvec2(E, l1), nv, s1);
    // strict '>' keeps the first candidate on ties, like algorithms.py:250; LAPACK's eigenvalue
    // ORDER is not reproduced (it only matters on exact ties)
    IaSolution s = (s1.capacity > s0.capacity) ? s1 : s0;
    s.ok = ok && (s.capacity == s.capacity);
    return s;
}
__device__ __noinline__ IaSolution ia_closed_form(const M2 (&H)[3][3], double nv) { return ia_closed_form_inl(H, nv); }

// ---- iterative solvers on the same 2x2 / one-stream geometry ----------------------
// Reference: ia/algorithms.py:802-883 (solve loop, _is_diff_significant :755-800), :1010-1129 (alternating
// minimisation), :1173-1240 (minimum leakage), :1265-1507 (max SINR); ia/iabase.py:600-667 (Q, Q_rev).
// Eigenvectors follow the LAPACK convention above, so the precoder phases -- which rotate the post-filter
// noise -- are the reference's.  The noise_var * I term the reference adds to Q (multiuser.py:1376-1380) only
// shifts the eigenvalues and is left out.
enum { IA_CLOSED_FORM = 0, IA_ALT_MIN = 1, IA_MIN_LEAKAGE = 2, IA_MAX_SINR = 3, IA_MMSE = 4 };

__device__ __forceinline__ M2 outer2(const V2& a) {  // a a^H
    M2 r;
    r.a = mk<double>(cabs2(a.x), 0.0);
    r.b = cmulc(a.x, a.y);   // a.x conj(a.y)
    r.c = cconj(r.b);
    r.d = mk<double>(cabs2(a.y), 0.0);
    return r;
}
__device__ __forceinline__ M2 madd2(const M2& p, const M2& q) {
    return M2{cadd(p.a, q.a), cadd(p.b, q.b), cadd(p.c, q.c), cadd(p.d, q.d)};
}
__device__ __forceinline__ M2 mherm(const M2& p) { return M2{cconj(p.a), cconj(p.c), cconj(p.b), cconj(p.d)}; }
__device__ __forceinline__ M2 mzero() {
    const cd z = mk<double>(0, 0);
    return M2{z, z, z, z};
}
// eigenvectors of a Hermitian 2x2: [0] smallest eigenvalue (leig), [1] largest (peig)
__device__ __forceinline__ void heig2(const M2& A, V2& v_small, V2& v_large) {
    const double a = A.a.x, d = A.d.x;
    const double half = 0.5 * (a - d);
    const double r = sqrt(half * half + cabs2(A.b));
    const double mid = 0.5 * (a + d);
    v_small = eigvec2(A, mk<double>(mid - r, 0.0));
    v_large = eigvec2(A, mk<double>(mid + r, 0.0));
}
__device__ __forceinline__ bool ia_diff_significant(const V2 (&Fo)[3], const V2 (&Fn)[3], double rel) {
    bool sig = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double mn = sqrt(fmin(cabs2(Fn[k].x), cabs2(Fn[k].y)));
        const double dx = sqrt(cabs2(csub(Fn[k].x, Fo[k].x))), dy = sqrt(cabs2(csub(Fn[k].y, Fo[k].y)));
        sig = sig || (fmax(dx, dy) > mn * rel);
    }
    return sig;
}

// MMSEIASolver._calc_Vi (algorithms.py:1660-1825) for a 2x2 Hermitian A = sum_k H_ki^H U_k U_k^H H_ki and
// b = H_ii^H U_i: V = (A + mu I)^-1 b with the smallest mu >= 0 giving |V|^2 <= P = 1.  The reference finds mu with
// scipy's secant iteration (tolerance 1.5e-8); here Newton on the secular function
// g(mu) = |c-|^2 / (l- + mu)^2 + |c+|^2 / (l+ + mu)^2 - 1 in the eigenbasis of A, convex and decreasing, so the
// iterates rise monotonically to the root from mu = 0.
__device__ __forceinline__ V2 ia_mmse_precoder(M2 A, const V2& b) {
    const double half = 0.5 * (A.a.x - A.d.x);
    const double r = sqrt(half * half + cabs2(A.b));
    const double mid = 0.5 * (A.a.x + A.d.x);
    double lm = mid - r, lp = mid + r;
    if (lp > 5e4 * lm) {                      // diagonal loading of an ill-conditioned sum (:1688-1692)
        const double load = 0.5 * (lm + lp) / 100.0;
        A.a.x += load;
        A.d.x += load;
        lm += load;
        lp += load;
    }
    V2 um, up;
    heig2(A, um, up);
    const double cm = cabs2(cadd(cmulc(b.x, um.x), cmulc(b.y, um.y)));   // |u-^H b|^2
    const double cp = cabs2(cadd(cmulc(b.x, up.x), cmulc(b.y, up.y)));
    double mu = 0.0;
    if (cm / (lm * lm) + cp / (lp * lp) > 1.0) {
        for (int it = 0; it < 100; ++it) {
            const double dm = lm + mu, dp = lp + mu;
            const double g = cm / (dm * dm) + cp / (dp * dp) - 1.0;
            const double gp = -2.0 * (cm / (dm * dm * dm) + cp / (dp * dp * dp));
            const double step = -g / gp;
            mu += step;
            if (fabs(step) <= 1e-16 * (mu + lp)) break;
        }
    }
    // V = adj(A + mu I) b / det(A + mu I)
    const cd a = mk<double>(A.a.x + mu, 0.0), d = mk<double>(A.d.x + mu, 0.0);
    const double det = a.x * d.x - cabs2(A.b);
    V2 v;
    v.x = cscale(csub(cmul(d, b.x), cmul(A.b, b.y)), 1.0 / det);
    v.y = cscale(csub(cmul(a, b.y), cmul(A.c, b.x)), 1.0 / det);
    return v;
}

// F: in = initial precoders (unit norm), out = solution (unit norm except for the MMSE solver, whose full_F
// carries the power constraint); Wh = rows W^H.  Returns the iterations run.
__device__ __forceinline__ int ia_iterate(const M2 (&H)[3][3], int algo, double nv, int max_iter, double rel,
                                          V2 (&F)[3], V2 (&Wh)[3], bool& ok, const V2* W_init = nullptr) {
    V2 W[3];    // alt-min: C_k (interference subspace); otherwise the receive vectors W_k
    V2 Fn[3];   // normalised precoders (what _is_diff_significant compares); F holds full_F
#pragma unroll
    for (int k = 0; k < 3; ++k) Fn[k] = F[k];
    auto interference = [&](int k, const V2 (&P)[3], bool reversed) {
        M2 Q = mzero();
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (l == k) continue;
            Q = madd2(Q, outer2(reversed ? mvec(mherm(H[l][k]), P[l]) : mvec(H[k][l], P[l])));
        }
        return Q;
    };
    auto update_W = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            M2 Q = interference(k, F, false);
            if (algo == IA_MMSE) {            // all K terms + sigma^2 I, not normalised (:1560-1600)
                Q = madd2(Q, outer2(mvec(H[k][k], F[k])));
                Q.a.x += nv;
                Q.d.x += nv;
                W[k] = mvec(minv(Q, ok), mvec(H[k][k], F[k]));
            } else if (algo == IA_MAX_SINR) {
                Q.a.x += nv;
                Q.d.x += nv;
                W[k] = vnormalize(mvec(minv(Q, ok), mvec(H[k][k], F[k])));
            } else {
                V2 lo, hi;
                heig2(Q, lo, hi);
                W[k] = algo == IA_ALT_MIN ? hi : lo;
            }
        }
    };
    auto update_F = [&]() {
        if (algo == IA_ALT_MIN) {
            M2 Y[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const M2 cc = outer2(W[k]);
                Y[k] = M2{mk<double>(1.0 - cc.a.x, 0.0), mk<double>(-cc.b.x, -cc.b.y), mk<double>(-cc.c.x, -cc.c.y),
                          mk<double>(1.0 - cc.d.x, 0.0)};
            }
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                M2 M = mzero();
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (k == l) continue;
                    M = madd2(M, mmul(mmul(mherm(H[k][l]), Y[k]), H[k][l]));
                }
                M.a.y = M.d.y = 0.0;             // Hermitian by construction
                M.c = cconj(M.b);
                V2 lo, hi;
                heig2(M, lo, hi);
                F[l] = lo;
            }
        } else {
            V2 Fx[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                M2 Q = interference(k, W, true);
                if (algo == IA_MMSE) {
                    const V2 b = mvec(mherm(H[k][k]), W[k]);
                    Q = madd2(Q, outer2(b));
                    Q.a.y = Q.d.y = 0.0;
                    Q.c = cconj(Q.b);
                    Fx[k] = ia_mmse_precoder(Q, b);
                } else if (algo == IA_MAX_SINR) {
                    Q.a.x += nv;
                    Q.d.x += nv;
                    Fx[k] = vnormalize(mvec(minv(Q, ok), mvec(mherm(H[k][k]), W[k])));
                } else {
                    V2 lo, hi;
                    heig2(Q, lo, hi);
                    Fx[k] = lo;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) F[k] = Fx[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) Fn[k] = algo == IA_MMSE ? vnormalize(F[k]) : F[k];
    };
    if (W_init != nullptr && algo != IA_ALT_MIN) {
#pragma unroll
        for (int k = 0; k < 3; ++k) W[k] = W_init[k];    // receive filters handed over by the initialisation
    } else {
        update_W();      // _before_initialize_W_func / _updateW on the initial precoder
    }
    int runned = 0;
    for (int it = 0; it < max_iter; ++it) {
        V2 Fo[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) Fo[k] = Fn[k];
        ++runned;
        update_F();
        update_W();
        if (!ia_diff_significant(Fo, Fn, rel)) break;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (algo == IA_ALT_MIN) {
            // first row of inv([H_kk F_k, C_k])
            const V2 a = mvec(H[k][k], F[k]);
            const cd det = csub(cmul(a.x, W[k].y), cmul(W[k].x, a.y));
            ok = ok && (cabs2(det) > 1e-280);
            Wh[k].x = cdiv_(W[k].y, det);
            Wh[k].y = cdiv_(mk<double>(-W[k].x.x, -W[k].x.y), det);
        } else {
            Wh[k].x = cconj(W[k].x);
            Wh[k].y = cconj(W[k].y);
        }
    }
    return runned;
}

// full_W_H (iabase.py:299-327), SINR (:768-789, :897-996) and sum capacity of a (F, W^H) pair, P = 1
__device__ __forceinline__ void ia_finish(const M2 (&H)[3][3], const V2 (&F)[3], const V2 (&Wh)[3], double nv,
                                          IaSolution& s) {
    s.capacity = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        s.F[k] = F[k];
        const V2 hf = mvec(H[k][k], F[k]);
        const cd eq = cadd(cmul(Wh[k].x, hf.x), cmul(Wh[k].y, hf.y));
        s.U[k].x = cdiv_(Wh[k].x, eq);
        s.U[k].y = cdiv_(Wh[k].y, eq);
        double den = nv * (cabs2(s.U[k].x) + cabs2(s.U[k].y));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j == k) continue;
            const V2 g = mvec(H[k][j], F[j]);
            den += cabs2(cadd(cmul(s.U[k].x, g.x), cmul(s.U[k].y, g.y)));
        }
        s.sinr[k] = 1.0 / den;
        s.capacity += log2(1.0 + s.sinr[k]);
    }
}

// init (IterativeIASolverBaseClass._solve_init, algorithms.py:633-663): IA_INIT_GIVEN = start from F_init
// ('random' / 'fix'); IA_INIT_CLOSED_FORM = F and W of the closed-form solution (:572-597); IA_INIT_ALT_MIN = run
// the alternating-minimisation solver from F_init with the same max_iterations and start from its F and its
// normalised receive filters (:599-632); IA_INIT_SVD = the most significant right singular vector of every user's
// direct channel (:503-547; unique up to a phase -- LAPACK's in the reference, the Hermitian eigen-solver's here).
enum { IA_INIT_GIVEN = 0, IA_INIT_CLOSED_FORM = 1, IA_INIT_ALT_MIN = 2, IA_INIT_SVD = 3 };

__device__ __noinline__ IaSolution ia_iterative(const M2 (&H)[3][3], int algo, double nv, int max_iter, double rel,
                                                int init, const V2 (&F_init)[3], int& runned) {
    V2 F[3], Wh[3], W0[3];
    const V2* W_init = nullptr;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) F[k] = F_init[k];
    if (init == IA_INIT_CLOSED_FORM) {
        const IaSolution c = ia_closed_form(H, nv);
        ok = c.ok;
#pragma unroll
        for (int k = 0; k < 3; ++k) F[k] = c.F[k];
        // W_k = leig of the interference at receiver k (rank one after alignment), LAPACK normalisation
        const V2 a[3] = {mvec(H[0][1], F[1]), mvec(H[1][0], F[0]), mvec(H[2][0], F[0])};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            V2 hi;
            heig2(outer2(a[k]), W0[k], hi);
        }
        W_init = W0;
    } else if (init == IA_INIT_SVD) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // H_kk^H H_kk = V diag(s^2) V^H: the eigenvector of the larger eigenvalue
            const M2& h = H[k][k];
            M2 g;
            g.a = mk<double>(cabs2(h.a) + cabs2(h.c), 0.0);
            g.d = mk<double>(cabs2(h.b) + cabs2(h.d), 0.0);
            g.b = cadd(cmul(cconj(h.a), h.b), cmul(cconj(h.c), h.d));
            g.c = cconj(g.b);
            V2 lo;
            heig2(g, lo, F[k]);
        }
    } else if (init == IA_INIT_ALT_MIN) {
        V2 Wha[3];
        ia_iterate(H, IA_ALT_MIN, nv, max_iter, rel, F, Wha, ok);
#pragma unroll
        for (int k = 0; k < 3; ++k) W0[k] = vnormalize(V2{cconj(Wha[k].x), cconj(Wha[k].y)});
        W_init = W0;
    }
    runned = ia_iterate(H, algo, nv, max_iter, rel, F, Wh, ok, W_init);
    IaSolution s;
    ia_finish(H, F, Wh, nv, s);
    s.ok = ok && (s.capacity == s.capacity);
    return s;
}

__device__ __forceinline__ void load_blocks(const cd* bigH, M2 (&H)[3][3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            H[k][l].a = bigH[(2 * k) * 6 + 2 * l];
            H[k][l].b = bigH[(2 * k) * 6 + 2 * l + 1];
            H[k][l].c = bigH[(2 * k + 1) * 6 + 2 * l];
            H[k][l].d = bigH[(2 * k + 1) * 6 + 2 * l + 1];
        }
}

// operator-level solver on injected channels: bigH [batch][6][6] (f64)
__global__ __launch_bounds__(64) void k_ia_closed_form(const cd* __restrict__ bigH, double nv, cd* __restrict__ F,
                                                       cd* __restrict__ U, double* __restrict__ sinr,
                                                       double* __restrict__ cap, uint32_t* __restrict__ skipped,
                                                       size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        M2 H[3][3];
        load_blocks(bigH + b * 36, H);
        const IaSolution s = ia_closed_form(H, nv);
        for (int k = 0; k < 3; ++k) {
            F[(b * 3 + k) * 2] = s.F[k].x;
            F[(b * 3 + k) * 2 + 1] = s.F[k].y;
            U[(b * 3 + k) * 2] = s.U[k].x;
            U[(b * 3 + k) * 2 + 1] = s.U[k].y;
            if (sinr) sinr[b * 3 + k] = s.sinr[k];
        }
        if (cap) cap[b] = s.capacity;
        if (skipped) skipped[b] = s.ok ? 0u : 1u;
    }
}

// operator-level iterative solver on injected channels and initial precoders: F_init [batch][3][2]
__global__ __launch_bounds__(64) void k_ia_iterative(const cd* __restrict__ bigH, const cd* __restrict__ F_init, int algo,
                                                     int init, double nv, int max_iter, double rel, cd* __restrict__ F,
                                                     cd* __restrict__ U, double* __restrict__ sinr,
                                                     double* __restrict__ cap, uint32_t* __restrict__ iters,
                                                     uint32_t* __restrict__ skipped, size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        M2 H[3][3];
        load_blocks(bigH + b * 36, H);
        V2 F0[3];
        for (int k = 0; k < 3; ++k) F0[k] = V2{F_init[(b * 3 + k) * 2], F_init[(b * 3 + k) * 2 + 1]};
        int runned = 0;
        const IaSolution s = ia_iterative(H, algo, nv, max_iter, rel, init, F0, runned);
        for (int k = 0; k < 3; ++k) {
            F[(b * 3 + k) * 2] = s.F[k].x;
            F[(b * 3 + k) * 2 + 1] = s.F[k].y;
            U[(b * 3 + k) * 2] = s.U[k].x;
            U[(b * 3 + k) * 2 + 1] = s.U[k].y;
            if (sinr) sinr[b * 3 + k] = s.sinr[k];
        }
        if (cap) cap[b] = s.capacity;
        if (iters) iters[b] = (uint32_t)runned;
        if (skipped) skipped[b] = s.ok ? 0u : 1u;
    }
}

// fused config 5: one wavefront per CHUNK of 64 realizations.  Phase 1: lane i solves realization i of the chunk
// (channel draw, solver -- closed form or iterative -- in f64 registers) and parks the end-to-end link in LDS:
// G[k][l] = U_k H_kl F_l (symbol of user l -> estimate of user k: the wanted gain on the diagonal, the residual
// interference off it) and the receive filters U_k.  Phase 2: the whole wave runs each realization's symbols,
// est_k = sum_l G_kl x_l + U_k . n_k, two columns per lane and pass (wave_draws.hpp).  With one solve per lane
// instead of the same solve on all 64 lanes the iterative solvers cost 1/64 of a wave per realization.
// Split in two launches since round 2: the per-lane solve wants every register the SIMD has (384 VGPRs + scratch in
// f64: one wave per SIMD), the symbol walk wants many resident waves and few registers.  Fused, the walk -- 98 % of the
// instructions -- ran at the solve's occupancy, VALU-busy 0.54; the record that crosses HBM between the two launches is
// 16 complex numbers per realization (G, U, flag).
constexpr int kIaRec = 16;      // G[9], U[6], {ok, 0}

// ITER = false: the closed-form solver only (config 5).  One kernel for both used to carry the iterative solvers' call frames
// (ia_iterative is __noinline__) into the closed-form launches too: 322 VGPRs + 66 AGPRs and 1 792 B of scratch per lane for a
// solve that needs neither.
template <typename T, bool ITER>
__global__ __launch_bounds__(64) void k_ia_solve_links(double noise_var, int solver, int init, int max_iter, double rel,
                                                       uint64_t seed, uint64_t first, uint64_t count,
                                                       cx<T>* __restrict__ recs, double* __restrict__ cap_out,
                                                       uint32_t* __restrict__ iter_out) {
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const Rng rng(seed, first + rl);
    cd bigH[36];
#pragma unroll
    for (int i = 0; i < 18; ++i)            // big_H row-major, two CN samples per Philox block
        cn_pair<double>(rng, STREAM_CHAN, (uint32_t)i, 1.0, bigH[2 * i], bigH[2 * i + 1]);
    M2 H[3][3];
    load_blocks(bigH, H);
    IaSolution s;
    int runned = 0;
    if (!ITER || solver == IA_CLOSED_FORM) {
        s = ia_closed_form_inl(H, noise_var);
    } else if constexpr (ITER) {
        // randomizeF (iabase.py:538-540): F_k = normalized(randn_c(Nt, Ns)) from the solver's own stream
        V2 F0[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            cd a, b;
            cn_pair<double>(rng, STREAM_PHASE, (uint32_t)k, 1.0, a, b);
            F0[k] = vnormalize(V2{a, b});
        }
        s = ia_iterative(H, solver, noise_var, max_iter, rel, init, F0, runned);
    }
    cx<T>* rec = recs + rl * kIaRec;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const V2 hf = mvec(H[k][l], s.F[l]);
            const cd g = cadd(cmul(s.U[k].x, hf.x), cmul(s.U[k].y, hf.y));
            rec[3 * k + l] = mk<T>((T)g.x, (T)g.y);
        }
        rec[9 + 2 * k] = mk<T>((T)s.U[k].x.x, (T)s.U[k].x.y);
        rec[9 + 2 * k + 1] = mk<T>((T)s.U[k].y.x, (T)s.U[k].y.y);
    }
    rec[15] = mk<T>(s.ok ? (T)1 : (T)0, (T)0);
    if (cap_out) cap_out[rl] = s.capacity;
    if (iter_out) iter_out[rl] = (uint32_t)runned;
}

// The symbol walk: one wavefront per `per_wave` consecutive realizations, est_k = sum_l G_kl x_l + U_k . n_k, two
// columns per lane and pass (wave_draws.hpp).  The record of a realization is wave-uniform (scalar loads).
template <typename T>
// (round 6: workgroups of FOUR independent wavefronts, one flush of the counters -- totals.hpp: wg_flush_waves)
__global__ __launch_bounds__(256, sizeof(T) == 4 ? MCLE_F32_WALK_WAVES : 3) void k_ia_link(ModemParams<T> mp, int n_symbols, double noise_var,
                                                                        uint64_t seed, uint64_t first, uint64_t count,
                                                                        int per_wave, const cx<T>* __restrict__ recs,
                                                                        mcle_counters* counters,
                                                                        uint32_t* __restrict__ sym_out,
                                                                        uint32_t* __restrict__ bit_out) {
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
    const bool packed = sizeof(T) == 4 && mp.method == MCLE_DEMOD_QAM_SLICER;
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
            const cx<T>* rec = recs + rl * kIaRec;
            cx<T> G[3][3], U[3][2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int l = 0; l < 3; ++l) G[k][l] = rec[3 * k + l];
                U[k][0] = rec[9 + 2 * k];
                U[k][1] = rec[9 + 2 * k + 1];
            }
            const bool ok = rec[15].x != (T)0;
            unsigned se = 0, be = 0;
            auto column = [&](const int (&tx)[3], const cx<T> (&nz)[6]) {
                cx<T> x[3], est[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) x[k] = s_table[tx[k]];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    est[k] = cmul(U[k][0], nz[2 * k]);
                    est[k] = cfma4(U[k][1], nz[2 * k + 1], est[k]);
#pragma unroll
                    for (int l = 0; l < 3; ++l) est[k] = cfma4(G[k][l], x[l], est[k]);
                }
                if constexpr (sizeof(T) == 4) {
                    if (packed) {   // the three decisions of the column in one packed level-domain slice (qam_pack.hpp)
                        const f4q er = {est[0].x, est[1].x, est[2].x, 0.f}, ei = {est[0].y, est[1].y, est[2].y, 0.f};
                        const uint32_t sent = (uint32_t)tx[0] | ((uint32_t)tx[1] << 8) | ((uint32_t)tx[2] << 16);
                        qam_count4((qam_levels4(er, ei, qp) ^ labels_to_levels(sent, qp)) & 0x00FFFFFFu, qp, se, be);
                        return;
                    }
                }
                int dec[3];
                bool done = false;
                if constexpr (sizeof(T) == 4) {
                    if (lockstep) {       // the three streams searched in lockstep (same decisions as demod_one)
                        if (mp.M <= 8) demod_multi_cert(mp, est, dec, [&](int (&d_)[3]) { demod_mindist_multi<3>(s_tab4, mp.M, est, d_); });
                        else demod_multi_cert(mp, est, dec, [&](int (&d_)[3]) { demod_grid4_multi<3>(s_tab4, s_grid, mp.grid, mp.M, est, d_); });
                        done = true;
                    }
                }
                if (!done) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) dec[k] = demod_one(mp, s_table, s_grid, est[k]);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned e = (unsigned)(tx[k] ^ dec[k]);
                    se += (e != 0u);
                    be += __popc(e);
                }
            };
            if ((n_symbols & 1) == 0) {
                for (int t0 = 0; t0 < n_symbols; t0 += kPairCols) {
                    const int t = t0 + 2 * lane;
                    int ta[3], tb[3];
                    wave_symbol_pairs<3>(rng, 3, (uint32_t)n_symbols, (uint32_t)t0, mask, lane, ta, tb);   // randint(0, M, [3, NSymbs])
                    if (t < n_symbols) {
                        if constexpr (sizeof(T) == 8) {
                            // complex128: receiver by receiver -- the noise of ONE receiver's two antennas (both columns), its
                            // estimate, its decisions -- instead of all six antennas' draws first: 16 live noise registers instead
                            // of 48, which is what lets the registers be bounded for three wavefronts per SIMD (round 5)
                            cx<T> xa[3], xb[3];
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                xa[k] = s_table[ta[k]];
                                xb[k] = s_table[tb[k]];
                            }
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                cx<T> za[2], zb[2];
#pragma unroll
                                for (int a = 0; a < 2; ++a)
                                    cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)(2 * k + a) * (uint32_t)n_symbols + (uint32_t)t) >> 1, sigma,
                                                za[a], zb[a], s_bm);
                                cx<T> ea = cmul(U[k][0], za[0]), eb = cmul(U[k][0], zb[0]);
                                ea = cfma4(U[k][1], za[1], ea);
                                eb = cfma4(U[k][1], zb[1], eb);
#pragma unroll
                                for (int l = 0; l < 3; ++l) {
                                    ea = cfma4(G[k][l], xa[l], ea);
                                    eb = cfma4(G[k][l], xb[l], eb);
                                }
                                const unsigned da = (unsigned)(ta[k] ^ demod_one(mp, s_table, s_grid, ea));
                                const unsigned db = (unsigned)(tb[k] ^ demod_one(mp, s_table, s_grid, eb));
                                se += (da != 0u) + (db != 0u);
                                be += __popc(da) + __popc(db);
                            }
                        } else {
                            cx<T> za[6], zb[6];
#pragma unroll
                            for (int a = 0; a < 6; ++a)
                                cn_pair_lds(rng, STREAM_NOISE, ((uint32_t)a * (uint32_t)n_symbols + (uint32_t)t) >> 1, sigma,
                                            za[a], zb[a], s_bm);
                            column(ta, za);
                            column(tb, zb);
                        }
                    }
                }
            } else {
                for (int t = lane; t < n_symbols; t += 64) {
                    int tx[3];
                    cx<T> nz[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) tx[k] = (int)symbol_at(rng, (uint64_t)k * n_symbols + t, mask);
#pragma unroll
                    for (int a = 0; a < 6; ++a) nz[a] = cn_sample<T>(rng, STREAM_NOISE, (uint64_t)a * n_symbols + t, sigma);
                    column(tx, nz);
                }
            }
            se = wave_sum_u32(se);
            be = wave_sum_u32(be);
            if (lane == 0) wg_account(totals, se, be, !ok, rl, sym_out, bit_out);
        }
    }
    wg_flush_waves<4>(totals_all, counters, 3ull * (unsigned long long)n_symbols, 3ull * (unsigned long long)n_symbols * mp.bits);
}

constexpr uint64_t kSolveSlice = 1ull << 20;   // realizations per solve + walk pair: bounds the record buffer (128 MB here)

template <typename T>
int run_ia_impl(mcle_ctx* ctx, const mcle_ia_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit, double* d_cap, uint32_t* d_iter) {
    int rc;
    void* recs = nullptr;
    const uint64_t slice = count < kSolveSlice ? count : kSolveSlice;
    if ((rc = ctx->scratch((size_t)slice * kIaRec * sizeof(cx<T>), &recs))) return rc;
    const ModemParams<T> mp = pipe_modem<T>(ctx, cfg->demod_method);
    const size_t lds = (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long);
    const int per_wave = 16;      // 4 ... 64 realizations per wavefront measured: 1.98-2.05e8 realizations/s, no trend
    for (uint64_t off = 0; off < count; off += slice) {
        const uint64_t n = count - off < slice ? count - off : slice;
        auto solve = cfg->solver == IA_CLOSED_FORM ? k_ia_solve_links<T, false> : k_ia_solve_links<T, true>;
        hipLaunchKernelGGL(solve, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, cfg->noise_var,
                           cfg->solver, cfg->initialize_with, cfg->max_iterations, cfg->relative_factor, seed, first + off, n,
                           (cx<T>*)recs, d_cap ? d_cap + off : nullptr, d_iter ? d_iter + off : nullptr);
        MCLE_LAUNCH_CHECK();
        const uint64_t chunks = (n + per_wave - 1) / per_wave;
        const uint64_t cap = (uint64_t)ctx->n_cu * (sizeof(T) == 4 ? 4 : 3);                    // workgroups of four wavefronts
        const unsigned grid = (unsigned)oversubscribed_grid(ctx, cap, (chunks + 3) / 4, 2);     // a chunk is 8-16 realizations; one chunk per workgroup measured 1.5 x slower
        bool walked = false;
        {
            // an even number of columns >= 128: the packed walk of walk_f64.hpp (round 6; complex64 since its last day)
            if (link_walk_f64_fits(cfg->n_symbols) && !ctx->opt[MCLE_OPT_WALK_LEGACY]) {
#ifdef MCLE_EXPERIMENTS
              if constexpr (sizeof(T) == 8) {
#define MCLE_IA_ABL(V_) case V_: launch_link_walk_f64<IaWalk, V_>(ctx, mp, cfg->n_symbols, cfg->noise_var, seed, first + off, n, (const double2*)recs, d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr); walked = true; break;
                switch ((int)ctx->opt[MCLE_OPT_F64_VARIANT]) { MCLE_IA_ABL(1) MCLE_IA_ABL(2) MCLE_IA_ABL(4) MCLE_IA_ABL(6) MCLE_IA_ABL(8) MCLE_IA_ABL(16) MCLE_IA_ABL(31) default: break; }
#undef MCLE_IA_ABL
              }
#endif
                if (!walked)
                    launch_link_walk<T, IaWalk>(ctx, mp, cfg->n_symbols, cfg->noise_var, seed, first + off, n, (const cx<T>*)recs,
                                                d_counters, d_sym ? d_sym + off : nullptr, d_bit ? d_bit + off : nullptr);
                walked = true;
            }
        }
        if (!walked)
            hipLaunchKernelGGL(k_ia_link<T>, dim3(grid), dim3(256), lds, ctx->stream, mp, cfg->n_symbols, cfg->noise_var, seed,
                               first + off, n, per_wave, (const cx<T>*)recs, d_counters, d_sym ? d_sym + off : nullptr,
                               d_bit ? d_bit + off : nullptr);
        MCLE_LAUNCH_CHECK();
    }
    return MCLE_OK;
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_ia_closed_form(mcle_ctx* ctx, const void* d_bigH, double noise_var, void* d_F, void* d_U, double* d_sinr,
                        double* d_capacity, uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && d_bigH != nullptr && d_F != nullptr && d_U != nullptr, "null argument");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    hipLaunchKernelGGL(k_ia_closed_form, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream,
                       (const double2*)d_bigH, noise_var, (double2*)d_F, (double2*)d_U, d_sinr, d_capacity, d_skipped,
                       batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_ia_iterative(mcle_ctx* ctx, int solver, int initialize_with, const void* d_bigH, const void* d_F_init,
                      double noise_var, int max_iterations, double relative_factor, void* d_F, void* d_U,
                      double* d_sinr, double* d_capacity, uint32_t* d_iterations, uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && d_bigH != nullptr && d_F_init != nullptr && d_F != nullptr && d_U != nullptr,
                 "null argument");
    MCLE_REQUIRE(solver >= MCLE_IA_ALT_MIN && solver <= MCLE_IA_MMSE, "solver must be one of the iterative MCLE_IA_*");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(max_iterations >= 1, "max_iterations must be positive");
    MCLE_REQUIRE(initialize_with >= MCLE_IA_INIT_GIVEN && initialize_with <= MCLE_IA_INIT_SVD,
                 "unknown initialisation %d", initialize_with);
    // AlternatingMinIASolver.initialize_with setter, algorithms.py:928-935
    MCLE_REQUIRE(!(solver == MCLE_IA_ALT_MIN && initialize_with == MCLE_IA_INIT_ALT_MIN),
                 "Can't use 'alt_min' initialization with 'AlternatingMinIASolver' class 'alt_min'");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    hipLaunchKernelGGL(k_ia_iterative, dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream,
                       (const double2*)d_bigH, (const double2*)d_F_init, solver, initialize_with, noise_var, max_iterations,
                       relative_factor, (double2*)d_F, (double2*)d_U, d_sinr, d_capacity, d_iterations, d_skipped,
                       batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_run_ia(mcle_ctx* ctx, int dtype, const mcle_ia_cfg* cfg, uint64_t seed, uint64_t first, uint64_t count,
                mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err, double* d_sum_capacity,
                uint32_t* d_iterations) {
    int rc = check_pipe(ctx, dtype, cfg ? cfg->demod_method : 0, cfg);
    if (rc) return rc;
    // ClosedFormIASolver.solve asserts K == 3 (algorithms.py:210); this kernel is its 2x2, Ns = 1 case
    MCLE_REQUIRE(cfg->K == 3, cfg->solver == MCLE_IA_CLOSED_FORM
                                  ? "The ClosedFormIASolver class only works in a MIMO-IC scenario with 3 users."
                                  : "fused IA pipeline supports K = 3 users");
    MCLE_REQUIRE(cfg->nr == 2 && cfg->nt == 2 && cfg->ns == 1, "fused IA pipeline supports Nr = Nt = 2, Ns = 1");
    MCLE_REQUIRE(cfg->n_symbols >= 1, "n_symbols must be positive");
    MCLE_REQUIRE(cfg->noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(cfg->solver >= MCLE_IA_CLOSED_FORM && cfg->solver <= MCLE_IA_MMSE, "unknown IA solver %d", cfg->solver);
    MCLE_REQUIRE(cfg->solver == MCLE_IA_CLOSED_FORM || cfg->max_iterations >= 1, "max_iterations must be positive");
    MCLE_REQUIRE(cfg->initialize_with >= MCLE_IA_INIT_GIVEN && cfg->initialize_with <= MCLE_IA_INIT_SVD,
                 "unknown initialisation %d", cfg->initialize_with);
    MCLE_REQUIRE(!(cfg->solver == MCLE_IA_ALT_MIN && cfg->initialize_with == MCLE_IA_INIT_ALT_MIN),
                 "Can't use 'alt_min' initialization with 'AlternatingMinIASolver' class 'alt_min'");
    MCLE_REQUIRE(count <= 0x7fffffffull, "at most 2^31-1 realizations per call");
    if (count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    return dtype == MCLE_F32 ? run_ia_impl<float>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err,
                                                  d_sum_capacity, d_iterations)
                             : run_ia_impl<double>(ctx, cfg, seed, first, count, d_counters, d_sym_err, d_bit_err,
                                                   d_sum_capacity, d_iterations);
}

}  // extern "C"
