// kernels_ia_general.hip -- iterative interference alignment for general geometries: K <= 4 users, Nr, Nt <= 6 antennas
// (two instantiations of ia_general_body.hpp: 4 x 4 matrices for the common case, 6 x 6 beyond), per-user stream counts, with
// the greedy and brute-force stream selection wrappers.
//
// Reference (paths relative to pyphysim/): ia/algorithms.py:271-883 IterativeIASolverBaseClass (solve loop,
// _is_diff_significant, initialize_with 'fix' / 'svd'), :885-1129 AlternatingMinIASolver, :1132-1240 MinLeakageIASolver,
// :1243-1507 MaxSinrIASolver, :1853-2075 GreedStreamIASolver, :2057-2260 BruteForceStreamIASolver; ia/iabase.py:188-327
// full_F / full_W_H, :600-667 calc_Q / calc_Q_rev, :768-996 calc_SINR; util/misc.py:161-255 peig / leig.
// kernels_ia.hip keeps the register-resident K = 3, 2x2, one-stream special case (the throughput path of config 5);
// here one lane solves one channel realization with small matrices of run-time size in private memory (f64).
//
// Eigenvectors are determined up to a phase; LAPACK's choice is not reproduced (ours: the component of largest
// modulus is real and positive).  Everything the link sees -- SINRs, sum capacity, decisions, and the iterates as
// subspaces -- does not depend on that choice; the element-wise convergence test (_is_diff_significant) can trip an
// iteration earlier or later than the reference's, so iteration counts are equal when the test is not what ends the
// loop (tests/golden/f3c_ia_general.npz pins both regimes).
#include "common.hpp"

namespace mcle {
#define MCLE_IAG_NS iag
#define MCLE_IAG_D 4
#include "ia_general_body.hpp"
#undef MCLE_IAG_NS
#undef MCLE_IAG_D
#define MCLE_IAG_NS iag6
#define MCLE_IAG_D 6
#include "ia_general_body.hpp"
#undef MCLE_IAG_NS
#undef MCLE_IAG_D
}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_ia_solve_general(mcle_ctx* ctx, const mcle_ia_general_cfg* cfg, const void* d_bigH, const void* d_F_init,
                          void* d_F, void* d_U, double* d_sinr, double* d_capacity, uint32_t* d_iterations,
                          int32_t* d_ns, uint32_t* d_skipped, double* d_every_capacity, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && cfg != nullptr && d_bigH != nullptr && d_F != nullptr && d_U != nullptr, "null argument");
    MCLE_REQUIRE(cfg->K >= 2 && cfg->K <= iag::KM, "K must be in [2, %d]", iag::KM);
    MCLE_REQUIRE(cfg->nr >= 1 && cfg->nr <= iag6::D && cfg->nt >= 1 && cfg->nt <= iag6::D,
                 "Nr and Nt must be in [1, %d]", iag6::D);
    const bool big = cfg->nr > iag::D || cfg->nt > iag::D;       // matrices (and the padded array layout) of 6 x 6 instead of 4 x 4
    MCLE_REQUIRE(cfg->solver >= 1 && cfg->solver <= 3, "solver must be alt_min (1), min_leakage (2) or max_sinr (3)");
    MCLE_REQUIRE(cfg->initialize_with == 0 || cfg->initialize_with == 3, "initialize_with must be 'fix' (0) or 'svd' (3)");
    MCLE_REQUIRE(cfg->initialize_with != 3 || cfg->nr == cfg->nt, "the 'svd' start is defined for Nr == Nt");
    MCLE_REQUIRE(cfg->initialize_with != 0 || d_F_init != nullptr || cfg->stream_selection == 2,
                 "The precoder must be manually set, since you specified the 'fix' initialize_with option.");
    MCLE_REQUIRE(cfg->stream_selection >= 0 && cfg->stream_selection <= 2, "stream_selection must be 0, 1 or 2");
    MCLE_REQUIRE(cfg->stream_selection != 2 || cfg->nr == cfg->nt, "brute-force selection starts from 'svd' (Nr == Nt)");
    MCLE_REQUIRE(cfg->max_iterations >= 1 && cfg->noise_var >= 0.0 && cfg->relative_factor >= 0.0, "bad solver settings");
    if (cfg->stream_selection == 2) {
        long long combos = 1;
        for (int k = 0; k < cfg->K; ++k) combos *= cfg->ns[k] > 0 ? cfg->ns[k] : 1;
        MCLE_REQUIRE(combos <= 256, "brute-force selection: %lld stream combinations exceed the 256 of d_every_capacity", combos);
    }
    for (int k = 0; k < cfg->K; ++k) {
        const int lim = cfg->nr < cfg->nt ? cfg->nr : cfg->nt;
        MCLE_REQUIRE(cfg->ns[k] >= 1 && cfg->ns[k] <= lim, "Ns[%d] = %d must be in [1, min(Nr, Nt) = %d]", k, cfg->ns[k], lim);
    }
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    if (big) {
        iag6::Params pp;
        pp.K = cfg->K;
        pp.nr = cfg->nr;
        pp.nt = cfg->nt;
        for (int k = 0; k < iag6::KM; ++k) pp.ns[k] = k < cfg->K ? cfg->ns[k] : 0;
        pp.solver = cfg->solver;
        pp.init = cfg->initialize_with;
        pp.max_iter = cfg->max_iterations;
        pp.select = cfg->stream_selection;
        pp.nv = cfg->noise_var;
        pp.rel = cfg->relative_factor;
        hipLaunchKernelGGL(iag6::k_ia_general, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, ctx->stream, pp,
                           (const double2*)d_bigH, (const double2*)d_F_init, (double2*)d_F, (double2*)d_U, d_sinr,
                           d_capacity, d_iterations, d_ns, d_skipped, d_every_capacity, batch);
        MCLE_LAUNCH_CHECK();
        return MCLE_OK;
    }
    iag::Params pp;
    pp.K = cfg->K;
    pp.nr = cfg->nr;
    pp.nt = cfg->nt;
    for (int k = 0; k < iag::KM; ++k) pp.ns[k] = k < cfg->K ? cfg->ns[k] : 0;
    pp.solver = cfg->solver;
    pp.init = cfg->initialize_with;
    pp.max_iter = cfg->max_iterations;
    pp.select = cfg->stream_selection;
    pp.nv = cfg->noise_var;
    pp.rel = cfg->relative_factor;
    hipLaunchKernelGGL(iag::k_ia_general, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, ctx->stream, pp,
                       (const double2*)d_bigH, (const double2*)d_F_init, (double2*)d_F, (double2*)d_U, d_sinr,
                       d_capacity, d_iterations, d_ns, d_skipped, d_every_capacity, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
