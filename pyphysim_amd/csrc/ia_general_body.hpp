// ia_general_body.hpp -- the body of kernels_ia_general.hip, included once per capacity: namespace MCLE_IAG_NS with matrices of
// MCLE_IAG_D x MCLE_IAG_D entries (4: the round-2 envelope, Nr, Nt <= 4; 6: the reference's own application geometry K = 3,
// Nr = 5, Nt = 3, Ns = 2 of apps/ia/IA_Results_NrxNt(Ns).py:130-133 and its neighbours).  No include guard, on purpose.
namespace MCLE_IAG_NS {


constexpr int D = MCLE_IAG_D;    // antennas / streams per user (the instantiation's capacity)
constexpr int KM = 4;   // users
typedef double2 cd;

struct Mat {
    cd a[D][D];
};

__device__ __forceinline__ cd c0() { return mk<double>(0.0, 0.0); }
__device__ __forceinline__ double abs2(cd z) { return z.x * z.x + z.y * z.y; }

__device__ void mzero(Mat& M) {
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) M.a[i][j] = c0();
}
// C = A (r x m) * B (m x c)
__device__ void mm(const Mat& A, const Mat& B, int r, int m, int c, Mat& C) {
    Mat T;
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            cd s = c0();
            for (int k = 0; k < m; ++k) s = cadd(s, cmul(A.a[i][k], B.a[k][j]));
            T.a[i][j] = s;
        }
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) C.a[i][j] = T.a[i][j];
}
// C = A^H (A is m x r) * B (m x c)
__device__ void mm_h(const Mat& A, const Mat& B, int r, int m, int c, Mat& C) {
    Mat T;
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) {
            cd s = c0();
            for (int k = 0; k < m; ++k) s = cadd(s, cmul(cconj(A.a[k][i]), B.a[k][j]));
            T.a[i][j] = s;
        }
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) C.a[i][j] = T.a[i][j];
}
// Q += A (n x c) A^H
__device__ void add_outer(Mat& Q, const Mat& A, int n, int c) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            cd s = c0();
            for (int k = 0; k < c; ++k) s = cadd(s, cmul(A.a[i][k], cconj(A.a[j][k])));
            Q.a[i][j] = cadd(Q.a[i][j], s);
        }
}
__device__ double fro(const Mat& A, int r, int c) {
    double s = 0.0;
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) s += abs2(A.a[i][j]);
    return sqrt(s);
}
__device__ void scale(Mat& A, int r, int c, double f) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) A.a[i][j] = cscale(A.a[i][j], f);
}

// Hermitian eigen-decomposition by cyclic complex Jacobi rotations: A (n x n, Hermitian) -> eigenvalues w ascending,
// eigenvectors in the columns of V (unit norm, largest component real positive).
__device__ void heig(int n, Mat A, double* w, Mat& V) {
    mzero(V);
    for (int i = 0; i < n; ++i) V.a[i][i] = mk<double>(1.0, 0.0);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A.a[i][i].x * A.a[i][i].x;
            for (int j = i + 1; j < n; ++j) off += abs2(A.a[i][j]);
        }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const cd apq = A.a[p][q];
                const double g = sqrt(abs2(apq));
                if (g < 1e-300) continue;
                // phase e = apq / |apq|; rotate the (p, q) plane so that the entry vanishes
                const cd e = cscale(apq, 1.0 / g);
                const double app = A.a[p][p].x, aqq = A.a[q][q].x;
                const double tau = (aqq - app) / (2.0 * g);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                // columns: A <- A J, J = [[c, s e], [-s conj(e), c]] acting on (p, q)
                for (int k = 0; k < n; ++k) {
                    const cd akp = A.a[k][p], akq = A.a[k][q];
                    A.a[k][p] = csub(cscale(akp, c), cscale(cmul(akq, cconj(e)), s));
                    A.a[k][q] = cadd(cscale(cmul(akp, e), s), cscale(akq, c));
                    const cd vkp = V.a[k][p], vkq = V.a[k][q];
                    V.a[k][p] = csub(cscale(vkp, c), cscale(cmul(vkq, cconj(e)), s));
                    V.a[k][q] = cadd(cscale(cmul(vkp, e), s), cscale(vkq, c));
                }
                // rows: A <- J^H A
                for (int k = 0; k < n; ++k) {
                    const cd apk = A.a[p][k], aqk = A.a[q][k];
                    A.a[p][k] = csub(cscale(apk, c), cscale(cmul(aqk, e), s));
                    A.a[q][k] = cadd(cscale(cmul(apk, cconj(e)), s), cscale(aqk, c));
                }
                A.a[p][q] = c0();
                A.a[q][p] = c0();
                A.a[p][p].y = 0.0;
                A.a[q][q].y = 0.0;
            }
    }
    for (int i = 0; i < n; ++i) w[i] = A.a[i][i].x;
    // sort ascending (selection sort on at most 4 values), carrying the columns
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j)
            if (w[j] < w[m]) m = j;
        if (m != i) {
            const double t = w[i];
            w[i] = w[m];
            w[m] = t;
            for (int k = 0; k < n; ++k) {
                const cd v = V.a[k][i];
                V.a[k][i] = V.a[k][m];
                V.a[k][m] = v;
            }
        }
    }
    // canonical phase per column
    for (int j = 0; j < n; ++j) {
        int m = 0;
        double best = -1.0;
        for (int k = 0; k < n; ++k) {
            const double a = abs2(V.a[k][j]);
            if (a > best * (1.0 + 1e-12)) {
                best = a;
                m = k;
            }
        }
        const double r = sqrt(best);
        if (r > 0.0) {
            const cd ph = cscale(cconj(V.a[m][j]), 1.0 / r);
            for (int k = 0; k < n; ++k) V.a[k][j] = cmul(V.a[k][j], ph);
            V.a[m][j].y = 0.0;
        }
    }
}
// eigenvectors of the m smallest eigenvalues, ascending (util/misc.py:210-255 leig)
__device__ void leig(int n, const Mat& A, int m, Mat& out) {
    double w[D];
    Mat V;
    heig(n, A, w, V);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out.a[i][j] = V.a[i][j];
}
// eigenvectors of the m largest eigenvalues, descending (util/misc.py:161-207 peig)
__device__ void peig(int n, const Mat& A, int m, Mat& out) {
    double w[D];
    Mat V;
    heig(n, A, w, V);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) out.a[i][j] = V.a[i][n - 1 - j];
}
// A (n x n) X = B (n x m), Gaussian elimination with partial pivoting; X overwrites B
__device__ bool solve(int n, Mat A, Mat& B, int m) {
    bool ok = true;
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = abs2(A.a[c][c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = abs2(A.a[r][c]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (best < 1e-280) {
            ok = false;
            continue;
        }
        if (p != c) {
            for (int k = 0; k < n; ++k) {
                const cd t = A.a[c][k];
                A.a[c][k] = A.a[p][k];
                A.a[p][k] = t;
            }
            for (int k = 0; k < m; ++k) {
                const cd t = B.a[c][k];
                B.a[c][k] = B.a[p][k];
                B.a[p][k] = t;
            }
        }
        const cd inv = cdivide(mk<double>(1.0, 0.0), A.a[c][c]);
        for (int r = c + 1; r < n; ++r) {
            const cd f = cmul(A.a[r][c], inv);
            for (int k = c; k < n; ++k) A.a[r][k] = csub(A.a[r][k], cmul(f, A.a[c][k]));
            for (int k = 0; k < m; ++k) B.a[r][k] = csub(B.a[r][k], cmul(f, B.a[c][k]));
        }
    }
    for (int k = 0; k < m; ++k)
        for (int r = n - 1; r >= 0; --r) {
            cd s = B.a[r][k];
            for (int c = r + 1; c < n; ++c) s = csub(s, cmul(A.a[r][c], B.a[c][k]));
            B.a[r][k] = abs2(A.a[r][r]) > 0.0 ? cdivide(s, A.a[r][r]) : c0();
        }
    return ok;
}

// ---- one realization ------------------------------------------------------------------------------------------
struct Problem {
    const cd* bigH;     // [K nr][K nt], row-major
    int K, nr, nt;
    double nv;
    __device__ void block(int k, int l, Mat& H) const {       // H_kl: receiver k, transmitter l
        for (int r = 0; r < nr; ++r)
            for (int c = 0; c < nt; ++c) H.a[r][c] = bigH[(size_t)(k * nr + r) * (K * nt) + l * nt + c];
    }
};
struct State {
    Mat F[KM];      // nt x ns[k]   (normalised precoders = full_F, P = 1)
    Mat W[KM];      // alt-min: C_k (nr x (nr - ns)); otherwise W_k (nr x ns)
    Mat WH[KM];     // ns x nr receive filters W^H after the solver finished
    int ns[KM];
};

// interference covariance at receiver k (+ nv I): iabase.py:600-640, multiuser.py:1345-1382
__device__ void calc_Q(const Problem& P, const State& S, const Mat* Fc, int k, bool noise, Mat& Q) {
    mzero(Q);
    Mat H, A;
    for (int l = 0; l < P.K; ++l) {
        if (l == k) continue;
        P.block(k, l, H);
        mm(H, Fc[l], P.nr, P.nt, S.ns[l], A);
        add_outer(Q, A, P.nr, S.ns[l]);
    }
    if (noise)
        for (int i = 0; i < P.nr; ++i) Q.a[i][i].x += P.nv;
}
// reverse network (iabase.py:642-667): sum_{l != k} (H_lk^H W_l)(H_lk^H W_l)^H, nt x nt
__device__ void calc_Q_rev(const Problem& P, const State& S, int k, Mat& Q) {
    mzero(Q);
    Mat H, A;
    for (int l = 0; l < P.K; ++l) {
        if (l == k) continue;
        P.block(l, k, H);
        mm_h(H, S.W[l], P.nt, P.nr, S.ns[l], A);
        add_outer(Q, A, P.nt, S.ns[l]);
    }
}

// Fc: the precoders the covariances are built from (full_F of the reference: iabase.py:600-640, 828-894).  They are
// S.F except in the first update of a greedy re-solve, where the wrapper has deleted a column from full_F without
// renormalising it while F was renormalised (algorithms.py:1962-1975).
__device__ void update_W(const Problem& P, State& S, int algo, const Mat* Fc) {
    Mat Q, H, A, Ac;
    for (int k = 0; k < P.K; ++k) {
        if (algo == 3) {     // max-SINR, per stream (algorithms.py:1376-1455)
            Mat first;
            mzero(first);
            for (int j = 0; j < P.K; ++j) {
                P.block(k, j, H);
                mm(H, Fc[j], P.nr, P.nt, S.ns[j], A);
                add_outer(first, A, P.nr, S.ns[j]);
            }
            P.block(k, k, H);
            mm(H, S.F[k], P.nr, P.nt, S.ns[k], A);            // directions H_kk v_l
            mm(H, Fc[k], P.nr, P.nt, S.ns[k], Ac);            // the same columns as the covariance sees them
            for (int l = 0; l < S.ns[k]; ++l) {
                Mat B = first, rhs;
                for (int i = 0; i < P.nr; ++i) {
                    for (int j = 0; j < P.nr; ++j) B.a[i][j] = csub(B.a[i][j], cmul(Ac.a[i][l], cconj(Ac.a[j][l])));
                    B.a[i][i].x += P.nv;
                    rhs.a[i][0] = A.a[i][l];
                }
                solve(P.nr, B, rhs, 1);
                double nrm = 0.0;
                for (int i = 0; i < P.nr; ++i) nrm += abs2(rhs.a[i][0]);
                nrm = 1.0 / sqrt(nrm);
                for (int i = 0; i < P.nr; ++i) S.W[k].a[i][l] = cscale(rhs.a[i][0], nrm);
            }
            scale(S.W[k], P.nr, S.ns[k], 1.0 / fro(S.W[k], P.nr, S.ns[k]));
        } else {
            calc_Q(P, S, Fc, k, true, Q);
            if (algo == 1) peig(P.nr, Q, P.nr - S.ns[k], S.W[k]);    // alt-min: C_k, the interference subspace
            else leig(P.nr, Q, S.ns[k], S.W[k]);                      // min leakage
        }
    }
}

__device__ void update_F(const Problem& P, State& S, int algo) {
    Mat Fn[KM], Q, H, A;
    if (algo == 1) {     // alternating minimisation (algorithms.py:1014-1058)
        Mat Y[KM];
        for (int k = 0; k < P.K; ++k) {
            const int nc = P.nr - S.ns[k];
            for (int i = 0; i < P.nr; ++i)
                for (int j = 0; j < P.nr; ++j) {
                    cd s = c0();
                    for (int c = 0; c < nc; ++c) s = cadd(s, cmul(S.W[k].a[i][c], cconj(S.W[k].a[j][c])));
                    Y[k].a[i][j] = mk<double>((i == j ? 1.0 : 0.0) - s.x, -s.y);
                }
        }
        for (int l = 0; l < P.K; ++l) {
            mzero(Q);
            for (int k = 0; k < P.K; ++k) {
                if (k == l) continue;
                P.block(k, l, H);
                mm(Y[k], H, P.nr, P.nr, P.nt, A);              // Y_k H_kl
                Mat T;
                mm_h(H, A, P.nt, P.nr, P.nt, T);               // H_kl^H Y_k H_kl
                for (int i = 0; i < P.nt; ++i)
                    for (int j = 0; j < P.nt; ++j) Q.a[i][j] = cadd(Q.a[i][j], T.a[i][j]);
            }
            for (int i = 0; i < P.nt; ++i) {                   // Hermitian by construction
                Q.a[i][i].y = 0.0;
                for (int j = i + 1; j < P.nt; ++j) Q.a[j][i] = cconj(Q.a[i][j]);
            }
            leig(P.nt, Q, S.ns[l], Fn[l]);
            scale(Fn[l], P.nt, S.ns[l], 1.0 / fro(Fn[l], P.nt, S.ns[l]));
        }
    } else if (algo == 2) {   // min leakage (algorithms.py:1180-1240)
        for (int k = 0; k < P.K; ++k) {
            calc_Q_rev(P, S, k, Q);
            leig(P.nt, Q, S.ns[k], Fn[k]);
        }
    } else {                  // max-SINR in the reverse network (algorithms.py:1265-1345, 1457-1480): every
        for (int k = 0; k < P.K; ++k) {   // "transmitter" j enters with power P / Ns_j per stream
            Mat first;
            mzero(first);
            for (int j = 0; j < P.K; ++j) {
                P.block(j, k, H);
                mm_h(H, S.W[j], P.nt, P.nr, S.ns[j], A);       // H_jk^H W_j
                scale(A, P.nt, S.ns[j], 1.0 / sqrt((double)S.ns[j]));
                add_outer(first, A, P.nt, S.ns[j]);
            }
            P.block(k, k, H);
            mm_h(H, S.W[k], P.nt, P.nr, S.ns[k], A);
            const double pk = 1.0 / (double)S.ns[k];
            for (int l = 0; l < S.ns[k]; ++l) {
                Mat B = first, rhs;
                for (int i = 0; i < P.nt; ++i) {
                    for (int j = 0; j < P.nt; ++j)
                        B.a[i][j] = csub(B.a[i][j], cscale(cmul(A.a[i][l], cconj(A.a[j][l])), pk));
                    B.a[i][i].x += P.nv;
                    rhs.a[i][0] = A.a[i][l];
                }
                solve(P.nt, B, rhs, 1);
                double nrm = 0.0;
                for (int i = 0; i < P.nt; ++i) nrm += abs2(rhs.a[i][0]);
                nrm = 1.0 / sqrt(nrm);
                for (int i = 0; i < P.nt; ++i) Fn[k].a[i][l] = cscale(rhs.a[i][0], nrm);
            }
            scale(Fn[k], P.nt, S.ns[k], 1.0 / fro(Fn[k], P.nt, S.ns[k]));
        }
    }
    for (int k = 0; k < P.K; ++k) S.F[k] = Fn[k];
}

// algorithms.py:700-760 _is_diff_significant
__device__ bool diff_significant(const Problem& P, const State& S, const Mat* Fo, double rel) {
    for (int k = 0; k < P.K; ++k) {
        double dmax = 0.0, fmin = 1e300;
        for (int i = 0; i < P.nt; ++i)
            for (int j = 0; j < S.ns[k]; ++j) {
                dmax = fmax(dmax, sqrt(abs2(csub(S.F[k].a[i][j], Fo[k].a[i][j]))));
                fmin = fmin < sqrt(abs2(S.F[k].a[i][j])) ? fmin : sqrt(abs2(S.F[k].a[i][j]));
            }
        if (dmax > fmin * rel) return true;
    }
    return false;
}

// util/misc.py:870-905 get_principal_component_matrix on the columns of A (rows x cols), keeping n components:
// A V_n (V_n[:n, :])^H with V_n the n dominant right singular vectors (sigma_i u_i = A v_i, so no division)
__device__ void principal_components(Mat& A, int rows, int cols, int n) {
    Mat G, V, T;
    double w[D];
    mm_h(A, A, cols, rows, cols, G);
    for (int i = 0; i < cols; ++i) {
        G.a[i][i].y = 0.0;
        for (int j = i + 1; j < cols; ++j) G.a[j][i] = cconj(G.a[i][j]);
    }
    heig(cols, G, w, V);                                       // ascending: dominant vector in column cols-1
    for (int i = 0; i < rows; ++i)
        for (int m = 0; m < n; ++m) {
            cd s = c0();
            for (int c = 0; c < cols; ++c) s = cadd(s, cmul(A.a[i][c], V.a[c][cols - 1 - m]));
            T.a[i][m] = s;
        }
    for (int i = 0; i < rows; ++i)
        for (int c = 0; c < n; ++c) {
            cd s = c0();
            for (int m = 0; m < n; ++m) s = cadd(s, cmul(T.a[i][m], cconj(V.a[c][cols - 1 - m])));
            A.a[i][c] = s;
        }
}

// algorithms.py:665-735 _solve_finalize: a multi-stream precoder whose condition number exceeds 1e4 has dead
// dimensions (an over-loaded allocation collapses onto fewer streams); keep the singular directions above max / 1e4
// in the precoder (renormalised) and in the receive filter, and lower the user's stream count.
__device__ void solve_finalize(const Problem& P, State& S) {
    for (int k = 0; k < P.K; ++k) {
        const int ns = S.ns[k];
        if (ns <= 1) continue;
        Mat G, V;
        double w[D];
        mm_h(S.F[k], S.F[k], ns, P.nt, ns, G);
        for (int i = 0; i < ns; ++i) {
            G.a[i][i].y = 0.0;
            for (int j = i + 1; j < ns; ++j) G.a[j][i] = cconj(G.a[i][j]);
        }
        heig(ns, G, w, V);                                     // squared singular values, ascending
        const double smax = sqrt(fmax(w[ns - 1], 0.0)), smin = sqrt(fmax(w[0], 0.0));
        if (!(smax > smin * 1e4)) continue;
        int n = 0;
        for (int i = 0; i < ns; ++i) n += sqrt(fmax(w[i], 0.0)) > smax / 1.0e4 ? 1 : 0;
        principal_components(S.F[k], P.nt, ns, n);
        scale(S.F[k], P.nt, n, 1.0 / fro(S.F[k], P.nt, n));
        Mat Wm;                                                // W = (W^H)^H, nr x ns
        for (int i = 0; i < P.nr; ++i)
            for (int j = 0; j < ns; ++j) Wm.a[i][j] = cconj(S.WH[k].a[j][i]);
        principal_components(Wm, P.nr, ns, n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < P.nr; ++j) S.WH[k].a[i][j] = cconj(Wm.a[j][i]);
        S.ns[k] = n;
    }
}

// IterativeIASolverBaseClass.solve from the precoders in S.F (initialize_with = 'fix'): -> runned iterations; S.WH set.
// Fc_first: precoders the FIRST receive-filter update builds its covariances from (null: S.F).
__device__ int run_solver(const Problem& P, State& S, int algo, int max_iter, double rel, bool& ok,
                          const Mat* Fc_first = nullptr) {
    update_W(P, S, algo, Fc_first ? Fc_first : S.F);
    int runned = 0;
    for (int it = 0; it < max_iter; ++it) {
        Mat Fo[KM];
        for (int k = 0; k < P.K; ++k) Fo[k] = S.F[k];
        ++runned;
        update_F(P, S, algo);
        update_W(P, S, algo, S.F);
        if (!diff_significant(P, S, Fo, rel)) break;
    }
    Mat H, A;
    for (int k = 0; k < P.K; ++k) {
        if (algo == 1) {      // W^H = first ns rows of inv([H_kk F_k, C_k]) (algorithms.py:1060-1090)
            P.block(k, k, H);
            mm(H, S.F[k], P.nr, P.nt, S.ns[k], A);
            Mat M, I;
            mzero(I);
            for (int i = 0; i < P.nr; ++i) {
                for (int j = 0; j < S.ns[k]; ++j) M.a[i][j] = A.a[i][j];
                for (int j = S.ns[k]; j < P.nr; ++j) M.a[i][j] = S.W[k].a[i][j - S.ns[k]];
                I.a[i][i] = mk<double>(1.0, 0.0);
            }
            ok = solve(P.nr, M, I, P.nr) && ok;
            for (int i = 0; i < S.ns[k]; ++i)
                for (int j = 0; j < P.nr; ++j) S.WH[k].a[i][j] = I.a[i][j];
        } else {
            for (int i = 0; i < S.ns[k]; ++i)
                for (int j = 0; j < P.nr; ++j) S.WH[k].a[i][j] = cconj(S.W[k].a[j][i]);
        }
    }
    solve_finalize(P, S);
    return runned;
}

// full_W_H (iabase.py:299-327), per-stream SINRs (:897-996) and sum capacity of (F, W^H)
__device__ double finish(const Problem& P, const State& S, Mat* U, double (*sinr)[D], bool& ok) {
    Mat H, A, E;
    double cap = 0.0;
    for (int k = 0; k < P.K; ++k) {
        P.block(k, k, H);
        mm(H, S.F[k], P.nr, P.nt, S.ns[k], A);                // H_kk F_k
        mm(S.WH[k], A, S.ns[k], P.nr, S.ns[k], E);            // W^H H_kk F_k
        U[k] = S.WH[k];
        ok = solve(S.ns[k], E, U[k], P.nr) && ok;             // U = (W^H H F)^-1 W^H
        Mat first, Hj, Aj;
        mzero(first);
        for (int j = 0; j < P.K; ++j) {
            P.block(k, j, Hj);
            mm(Hj, S.F[j], P.nr, P.nt, S.ns[j], Aj);
            add_outer(first, Aj, P.nr, S.ns[j]);
        }
        for (int l = 0; l < S.ns[k]; ++l) {
            cd num = c0();
            for (int i = 0; i < P.nr; ++i) num = cadd(num, cmul(U[k].a[l][i], A.a[i][l]));
            double den = 0.0;                                 // u^H B u with B = first - a a^H + nv I
            for (int i = 0; i < P.nr; ++i)
                for (int j = 0; j < P.nr; ++j) {
                    cd b = csub(first.a[i][j], cmul(A.a[i][l], cconj(A.a[j][l])));
                    if (i == j) b.x += P.nv;
                    den += cmul(cmul(U[k].a[l][i], b), cconj(U[k].a[l][j])).x;
                }
            sinr[k][l] = fabs(abs2(num) / den);
            cap += log2(1.0 + sinr[k][l]);
        }
    }
    return cap;
}

// initialize_with = 'svd' (algorithms.py:503-547): the ns most significant right singular vectors of H_kk, in the
// column order of least_right_singular_vectors' reversed index list (misc.py:647-660), Frobenius-normalised
__device__ void init_svd(const Problem& P, State& S) {
    Mat H, G, V;
    double w[D];
    for (int k = 0; k < P.K; ++k) {
        P.block(k, k, H);
        mm_h(H, H, P.nt, P.nr, P.nt, G);                      // H^H H
        for (int i = 0; i < P.nt; ++i) {
            G.a[i][i].y = 0.0;
            for (int j = i + 1; j < P.nt; ++j) G.a[j][i] = cconj(G.a[i][j]);
        }
        heig(P.nt, G, w, V);                                  // ascending: column nt-1 = dominant singular vector
        for (int j = 0; j < S.ns[k]; ++j) {                   // columns ns-1 ... 0 of the descending ordering
            const int desc = S.ns[k] - 1 - j;
            for (int i = 0; i < P.nt; ++i) S.F[k].a[i][j] = V.a[i][P.nt - 1 - desc];
        }
        scale(S.F[k], P.nt, S.ns[k], 1.0 / fro(S.F[k], P.nt, S.ns[k]));
    }
}

struct Params {
    int K, nr, nt;
    int ns[KM];
    int solver, init, max_iter, select;      // select: 0 none, 1 greedy, 2 brute force
    double nv, rel;
};

__global__ __launch_bounds__(64) void k_ia_general(Params pp, const cd* __restrict__ bigH, const cd* __restrict__ F_init,
                                                   cd* __restrict__ F_out, cd* __restrict__ U_out,
                                                   double* __restrict__ sinr_out, double* __restrict__ cap_out,
                                                   uint32_t* __restrict__ iters_out, int32_t* __restrict__ ns_out,
                                                   uint32_t* __restrict__ skipped, double* __restrict__ combo_cap,
                                                   size_t batch) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    Problem P{bigH + b * (size_t)(pp.K * pp.nr) * (pp.K * pp.nt), pp.K, pp.nr, pp.nt, pp.nv};
    State S, best;
    Mat U[KM];
    double sinr[KM][D];
    bool ok = true;
    unsigned iters = 0;
    double cap = 0.0;
    auto load_init = [&](const int* ns) {
        for (int k = 0; k < P.K; ++k) {
            S.ns[k] = ns[k];
            if (pp.init == 0)
                for (int i = 0; i < P.nt; ++i)
                    for (int j = 0; j < ns[k]; ++j)
                        S.F[k].a[i][j] = F_init[((b * KM + k) * D + i) * D + j];
        }
        if (pp.init != 0) init_svd(P, S);
    };
    if (pp.select == 2) {                 // brute force over every stream combination, 'svd' start (:2147-2260)
        int comb[KM];
        for (int k = 0; k < P.K; ++k) comb[k] = 1;
        double best_cap = -1.0;
        bool first = true;
        int n_comb = 0;
        while (true) {
            for (int k = 0; k < P.K; ++k) S.ns[k] = comb[k];
            init_svd(P, S);
            bool okc = true;
            iters += (unsigned)run_solver(P, S, pp.solver, pp.max_iter, pp.rel, okc);
            const double c = finish(P, S, U, sinr, okc);
            if (combo_cap) combo_cap[b * 256 + n_comb] = c;       // every_sum_capacity, in stream_combinations order
            ++n_comb;
            if (first || c > best_cap) {
                best_cap = c;
                best = S;
                ok = okc;
                first = false;
            }
            int k = P.K - 1;                                  // itertools.product order: last user fastest
            while (k >= 0 && comb[k] == pp.ns[k]) comb[k--] = 1;
            if (k < 0) break;
            ++comb[k];
        }
        S = best;
        cap = finish(P, S, U, sinr, ok);
    } else {
        load_init(pp.ns);
        iters += (unsigned)run_solver(P, S, pp.solver, pp.max_iter, pp.rel, ok);
        cap = finish(P, S, U, sinr, ok);
        if (pp.select == 1) {             // greedy stream reduction (:1905-2010)
            // the wrapper re-solves with initialize_with = 'fix', which does not reset the solver's iteration
            // counter, and adds the solver's cumulative return values (:1936, :1979)
            unsigned counter = iters;
            bool any = false;
            for (int k = 0; k < P.K; ++k) any = any || S.ns[k] > 1;
            while (any) {
                best = S;
                const double old_cap = cap;
                // user with the worst per-user minimum SINR among users with more than one stream
                int user = -1, stream = 0;
                double worst = 1e300;
                for (int k = 0; k < P.K; ++k) {
                    if (S.ns[k] <= 1) continue;
                    int ml = 0;
                    for (int l = 1; l < S.ns[k]; ++l)
                        if (sinr[k][l] < sinr[k][ml]) ml = l;
                    if (sinr[k][ml] < worst) {
                        worst = sinr[k][ml];
                        user = k;
                        stream = ml;
                    }
                }
                for (int i = 0; i < P.nt; ++i)
                    for (int j = stream; j < S.ns[user] - 1; ++j) S.F[user].a[i][j] = S.F[user].a[i][j + 1];
                --S.ns[user];
                Mat Fc[KM];                // full_F: the column is gone, the norm is not restored (:1962-1967)
                for (int k = 0; k < P.K; ++k) Fc[k] = S.F[k];
                scale(S.F[user], P.nt, S.ns[user], 1.0 / fro(S.F[user], P.nt, S.ns[user]));
                bool okc = true;
                counter += (unsigned)run_solver(P, S, pp.solver, pp.max_iter, pp.rel, okc, Fc);
                iters += counter;
                cap = finish(P, S, U, sinr, okc);
                if (old_cap > cap) {
                    S = best;
                    cap = finish(P, S, U, sinr, ok);
                    break;
                }
                ok = ok && okc;
                any = false;
                for (int k = 0; k < P.K; ++k) any = any || S.ns[k] > 1;
            }
        }
    }
    for (int k = 0; k < KM; ++k) {
        const bool live = k < P.K;
        if (ns_out) ns_out[b * KM + k] = live ? S.ns[k] : 0;
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                F_out[((b * KM + k) * D + i) * D + j] = (live && i < P.nt && j < S.ns[k]) ? S.F[k].a[i][j] : c0();
                U_out[((b * KM + k) * D + i) * D + j] = (live && i < S.ns[k] && j < P.nr) ? U[k].a[i][j] : c0();
            }
        if (sinr_out)
            for (int l = 0; l < D; ++l) sinr_out[(b * KM + k) * D + l] = (live && l < S.ns[k]) ? sinr[k][l] : 0.0;
    }
    if (cap_out) cap_out[b] = cap;
    if (iters_out) iters_out[b] = iters;
    if (skipped) skipped[b] = ok ? 0u : 1u;
}

}  // namespace MCLE_IAG_NS
