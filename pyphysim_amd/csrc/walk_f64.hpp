// walk_f64.hpp -- the complex128 symbol walk of the link pipelines (config 5: kernels_ia.hip, f6: kernels_bd.hip) as ONE kernel
// body, round 6.  Stands in for the per-realization loop of the reference's applications -- the transmit symbols, the noise, the
// receive filter and the decisions of ONE channel realization (apps/ia/simulate_ia.py:94-245, apps/comp/simulate_comp.py with
// comm/blockdiagonalization.py:272-566; MultiUserChannelMatrix.corrupt_data, channels/multiuser.py:1179-1262) -- after the
// per-lane solve kernels left a record (the effective gains and the receive filters) per realization.
//
// What the section tables of the round-5 walks showed (profiles/r06/{c5,f6}_section_table.md; k_ia_link<double> / k_bd_link<double>):
//   * 17 - 20 % of their vector instructions were v_readlane_b32: the record (60 - 72 scalar registers), the modem parameters of
//     a run-time demod_one inlined per decision and the Box-Muller constants did not fit the scalar file -- 146 - 178 spilled SGPRs,
//     every use behind a read-back;
//   * a decision cost 75 - 85 instructions where the certificate needs ~10 - 25: every demod_one site carried the method switch, the
//     certificate switch, the candidate-grid search and the sweep, with the spilled parameters re-read around each branch;
//   * config 5's 200 columns are 100 lane pairs: a pass of 64 and a pass of 36 per realization, 22 % of the lanes idle.
// Here:
//   * the decision form is a template parameter (slicer, QAM margin certificate, the quadrant certificate and its on-axis twin
//     for the reference's PSK(4), or the generic demod_one);
//     the certificates of a user's decisions run straight-line and ONE guarded table sweep serves the lanes holding an
//     uncertified symbol (the literal first-minimum sweep the certificate stands for; ~1e-8 per symbol);
//   * the records of a chunk of realizations are staged in LDS once per chunk; a coefficient is a ds_read_b128 at its use;
//   * the lane pairs of a chunk are ONE index space p = realization-in-chunk * (n_symbols / 2) + pair: a pass is 64 consecutive
//     p whatever realization they belong to (at most two: n_symbols >= 128), so only the chunk's last pass can have idle lanes --
//     16 realizations of 200 columns are 25 full passes instead of 32.  The Philox counter carries the realization per lane
//     (the key is the seed), the record pointer is per lane, the per-realization error counts are two masked wave sums per pass;
//   * symbols: the <= 9 DATA blocks under each (realization segment, stream) run of a pass are evaluated by 9 lanes each in one
//     Philox call (two calls for more than 3 streams), stored as 16 bytes in LDS, and a lane reads its two labels as ONE 16-bit
//     word -- instead of four ds_bpermute and a select tree per stream.
// Draws and arithmetic are position by position those of the round-5 walks (wave_draws.hpp: symbol n = byte n & 15 of DATA
// block n >> 4, the two columns of a lane = the two samples of one NOISE block; estimates in the same association), so the counts
// are identical; tests/test_gpu_oracle_depth.py, test_gpu_walk_f64.py.
#pragma once
#include <cmath>
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"
#include "qam_pack.hpp"
#include "wave_draws.hpp"

namespace mcle {

enum : int { WDEC_GENERIC = 0, WDEC_SLICER = 1, WDEC_QAM_CERT = 2, WDEC_QUAD_CERT = 3, WDEC_AXIS4_CERT = 4 };

// host: the decision form a launch with these modem parameters compiles to (and, for the on-axis four-point form -- the
// reference's PSK(4), which has no entry in ModemParams::cert -- its constants in the quadrant certificate's fields)
template <typename T> inline int walk_dec_kind(const mcle_ctx* ctx, ModemParams<T>& mp) {
    if (mp.method == MCLE_DEMOD_QAM_SLICER) return WDEC_SLICER;
    if (mp.cert == 1) return WDEC_QAM_CERT;
    if (mp.cert == 2) return WDEC_QUAD_CERT;
    if (mp.method == MCLE_DEMOD_MINDIST && ctx->axis_ok && !ctx->opt[MCLE_OPT_DEMOD_NOCERT]) {
        // complex64: margin 2 a lo = 2^-14 a^2 against a rounding of <= (9 a)^2 2^-23 = 2^-16.7 a^2 of either metric inside |re|, |im| <= 8 a
        mp.quad_lut = ctx->axis_lut;
        mp.quad_lo = (T)(ctx->axis_a * (sizeof(T) == 8 ? 0x1p-30 : 0x1p-15));
        mp.quad_hi = (T)(ctx->axis_a * (sizeof(T) == 8 ? 0x1p+8 : 8.0));
        return WDEC_AXIS4_CERT;
    }
    return WDEC_GENERIC;
}

// A wavefront's own LDS instructions execute in order: a store followed by another lane's load needs no wait, only the compiler
// has to keep the program order (cf. pipeline_mimo_qw.hip).
__device__ __forceinline__ void walk_wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum of a 32-bit value over the wavefront WITHOUT the LDS crossbar: a shift-and-add scan inside the rows of sixteen lanes
// (row_shr 1, 2, 4, 8; lanes without a source add 0), then row_bcast:15 / row_bcast:31 carry the row totals up -- six v_add_u32 with DPP
// modifiers, the total in lane 63, read back as a scalar.  (wave_sum_u32 = six ds_bpermute_b32 round trips; twice per pass they were
// most of the walk's skeleton: profiles/r06/walk_f64_ab.log.)
__device__ __forceinline__ uint32_t walk_wave_total(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8: lane 15 of a row = the row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);      // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);      // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// the literal sweep (strict '<': numpy.argmin's first minimum, fundamental.py:241-246), rolled: it runs once in ~1e8 symbols
template <typename T> __device__ __forceinline__ int walk_sweep(const cx<T>* __restrict__ s_table, int M, cx<T> r) {
    T best = (r.x - s_table[0].x) * (r.x - s_table[0].x) + (r.y - s_table[0].y) * (r.y - s_table[0].y);
    int idx = 0;
#pragma unroll 1
    for (int m = 1; m < M; ++m) {
        const cx<T> c = s_table[m];
        const T dx = r.x - c.x, dy = r.y - c.y;
        const T d = dx * dx + dy * dy;
        if (d < best) {
            best = d;
            idx = m;
        }
    }
    return idx;
}

// demod_qam_slicer<double> (modem.hpp) with the clamp BEFORE the floor (v_max_f64 / v_min_f64: floor and an integer-bounded clamp
// commute; NaN -> 0 as before) instead of two compares and four selects per axis after it, and both Gray decodes in one register
// (a byte each, as demod_qam_cert does): ~24 instead of ~36 instructions per decision (profiles/r06/c5_section_table.md).  Same
// level arithmetic, same labels.
__device__ __forceinline__ int walk_qam_slicer(float2 r, float scale, int L, int half_bits) {     // (never reached: complex64 packs)
    return demod_qam_slicer<float>(r, scale, L, half_bits);
}
__device__ __forceinline__ int walk_qam_slicer(double2 r, double scale, int L, int half_bits) {
    const double lm1 = (double)(L - 1);
    const double tj = (r.x * scale + lm1) * 0.5 + 0.5, ti = (lm1 - r.y * scale) * 0.5 + 0.5;
    const int cj = (int)floor(fmin(fmax(tj, 0.0), lm1)), ci = (int)floor(fmin(fmax(ti, 0.0), lm1));
    unsigned v = ((unsigned)ci << 8) | (unsigned)cj;
    v ^= (v >> 4) & 0x0F0Fu;
    v ^= (v >> 2) & 0x3F3Fu;
    v ^= (v >> 1) & 0x7F7Fu;
    return (int)(((v >> 8) << half_bits) | (v & 0xFFu));
}

// complex128, square QAM, FOUR decisions counted in the LEVEL domain (the complex64 pipelines' trick, qam_pack.hpp): the decided levels
// (row << hb | column, a byte per symbol) are compared with the sent labels turned into levels (one shift-and-xor for all four), and
// the bit errors follow from one field-wise prefix xor of the difference word -- no Gray decode and no label per decision: ~18
// instead of ~27 instructions per decision.  CERT: demod_qam_cert's margin test (same t, same clamp, same rint, same bound); a symbol
// it does not vouch for takes the literal sweep, whose label goes back to the level domain.  !CERT: the slicer's floor(t + 1/2).
// `sent`: the four labels, a byte each.
template <typename T, bool CERT>
__device__ __forceinline__ void walk_qam_count4(const ModemParams<T>& mp, const cx<T>* __restrict__ s_table, const cx<T> (&e)[4],
                                                uint32_t sent, unsigned& se, unsigned& be) {
    static_assert(sizeof(T) == 8 || CERT, "complex64 slices in the packed form (qam_levels4)");
    const int hb = mp.half_bits;
    const uint32_t fm = (1u << hb) - 1u;
    QamPack qp;
    qp.hb = hb;
    qp.m1 = (((fm >> 1) | ((fm >> 1) << hb)) & 0xFFu) * 0x01010101u;
    qp.m2 = (((fm >> 2) | ((fm >> 2) << hb)) & 0xFFu) * 0x01010101u;
    const T lm1 = (T)(mp.qam_L - 1), hs = mp.qam_scale * (T)0.5, hl = lm1 * (T)0.5;
    constexpr T lim = sizeof(T) == 8 ? (T)(0.5 - 0x1p-30) : (T)(0.5 - 0x1p-15);      // demod_qam_cert's margins
    [[maybe_unused]] const T rmax = ((T)16 + hl) / hs;                      // complex64: the certificate's range (wave-uniform)
    uint32_t lv = 0u;
    [[maybe_unused]] bool sure[4], all = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int kj, ki;
        if constexpr (CERT) {                                       // modem.hpp: demod_qam_cert, operation for operation
            T tj = e[i].x * hs + hl, ti = hl - e[i].y * hs;
            tj = fmin(fmax(tj, (T)0), lm1);
            ti = fmin(fmax(ti, (T)0), lm1);
            const T rj = rint(tj), ri = rint(ti);
            sure[i] = fabs(tj - rj) <= lim && fabs(ti - ri) <= lim;
            if constexpr (sizeof(T) == 4) sure[i] = sure[i] && fabs(e[i].x) <= rmax && fabs(e[i].y) <= rmax;
            all = all && sure[i];
            kj = (int)rj;
            ki = (int)ri;
        } else {                                                    // walk_qam_slicer's levels
            const T tj = (e[i].x * mp.qam_scale + lm1) * (T)0.5 + (T)0.5, ti = (lm1 - e[i].y * mp.qam_scale) * (T)0.5 + (T)0.5;
            kj = (int)floor(fmin(fmax(tj, (T)0), lm1));
            ki = (int)floor(fmin(fmax(ti, (T)0), lm1));
        }
        lv |= (((uint32_t)ki << hb) | (uint32_t)kj) << (8 * i);
    }
    if constexpr (CERT) {
        if (!all) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (!sure[i]) {
                    const uint32_t lab = (uint32_t)walk_sweep<T>(s_table, mp.M, e[i]);
                    const uint32_t lev = lab ^ ((lab >> 1) & (qp.m1 & 0xFFu));
                    lv = (lv & ~(0xFFu << (8 * i))) | (lev << (8 * i));
                }
        }
    }
    qam_count4(lv ^ labels_to_levels(sent, qp), qp, se, be);
}
// (the complex128 callers that named only the form)
template <bool CERT>
__device__ __forceinline__ void walk_qam_count4(const ModemParams<double>& mp, const double2* __restrict__ s_table, const double2 (&e)[4],
                                                uint32_t sent, unsigned& se, unsigned& be) {
    walk_qam_count4<double, CERT>(mp, s_table, e, sent, se, be);
}

// N decisions and their error counts.  DEC fixes the form at compile time; the certificates are those of modem.hpp.  complex64:
// the slicer is the packed level-domain form of the complex64 pipelines (qam_pack.hpp: four decisions per v_cvt_pk_u8_f32 word).
template <typename T, int DEC, int N>
__device__ __forceinline__ void walk_decide(const ModemParams<T>& mp, const cx<T>* __restrict__ s_table,
                                            const unsigned long long* __restrict__ s_grid, const cx<T> (&e)[N],
                                            const int (&tx)[N], unsigned& se, unsigned& be) {
    if constexpr (DEC == WDEC_SLICER && sizeof(T) == 4) {
        const QamPack qp = qam_pack(mp);
#pragma unroll
        for (int g0 = 0; g0 < N; g0 += 4) {
            f4q er = {0.f, 0.f, 0.f, 0.f}, ei = {0.f, 0.f, 0.f, 0.f};
            uint32_t sent = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (g0 + i < N) {
                    er[i] = e[g0 + i].x;
                    ei[i] = e[g0 + i].y;
                    sent |= (uint32_t)tx[g0 + i] << (8 * i);
                }
            const uint32_t live = N - g0 >= 4 ? 0xFFFFFFFFu : ((1u << (8 * ((N - g0) & 3))) - 1u);     // (folds: the loop is unrolled)
            qam_count4((qam_levels4(er, ei, qp) ^ labels_to_levels(sent, qp)) & live, qp, se, be);
        }
        return;
    }
    // level-domain counting, four at a time: complex128 both forms; complex64 the margin certificate (last day of round 6: the full-wave
    // kernel's min-distance rate was 0.84 of its slicer rate, a label and a Gray decode per decision)
    if constexpr ((sizeof(T) == 8 ? (DEC == WDEC_SLICER || DEC == WDEC_QAM_CERT) : DEC == WDEC_QAM_CERT) && N % 4 == 0) {
#pragma unroll
        for (int g0 = 0; g0 < N; g0 += 4) {
            const cx<T> e4[4] = {e[g0], e[g0 + 1], e[g0 + 2], e[g0 + 3]};
            const uint32_t sent = (uint32_t)tx[g0] | ((uint32_t)tx[g0 + 1] << 8) | ((uint32_t)tx[g0 + 2] << 16) | ((uint32_t)tx[g0 + 3] << 24);
            walk_qam_count4<T, DEC == WDEC_QAM_CERT>(mp, s_table, e4, sent, se, be);
        }
        return;
    }
    int dec[N];
    if constexpr (DEC == WDEC_SLICER) {
#pragma unroll
        for (int j = 0; j < N; ++j) dec[j] = walk_qam_slicer(e[j], mp.qam_scale, mp.qam_L, mp.half_bits);
    } else if constexpr (DEC == WDEC_QAM_CERT || DEC == WDEC_QUAD_CERT || DEC == WDEC_AXIS4_CERT) {
        bool sure[N], all = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if constexpr (DEC == WDEC_QAM_CERT) dec[j] = demod_qam_cert<T>(e[j], mp.qam_scale, mp.qam_L, mp.half_bits, sure[j]);
            else if constexpr (DEC == WDEC_QUAD_CERT) dec[j] = demod_quad_cert<T>(e[j], mp.quad_lut, mp.quad_lo, mp.quad_hi, sure[j]);
            else dec[j] = demod_axis4_cert<T>(e[j], mp.quad_lut, mp.quad_lo, mp.quad_hi, sure[j]);
            all = all && sure[j];
        }
        if (!all) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                if (!sure[j]) dec[j] = walk_sweep<T>(s_table, mp.M, e[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) dec[j] = demod_one(mp, s_table, s_grid, e[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const unsigned x = (unsigned)(tx[j] ^ dec[j]);
        se += (x != 0u);
        be += __popc(x);
    }
}
// (the complex128 kernels of config 4's family call the complex128 form without naming the arithmetic)
template <int DEC, int N>
__device__ __forceinline__ void walk_decide(const ModemParams<double>& mp, const double2* __restrict__ s_table,
                                            const unsigned long long* __restrict__ s_grid, const double2 (&e)[N],
                                            const int (&tx)[N], unsigned& se, unsigned& be) {
    walk_decide<double, DEC, N>(mp, s_table, s_grid, e, tx, se, be);
}

// ---- the two link shapes ------------------------------------------------------------------------------------------------------
// K users; user k has R receive antennas (noise rows k R + a) and J streams (symbol rows k J + jj).  A record is STRIDE complex128
// values with the validity flag at OK_AT; gain(s, l) / mix(s, a) are the record positions of the coefficient of symbol row l and
// of noise antenna a in the estimate of stream s.
struct IaWalk {       // config 5: est_k = U_k0 n_k0 + U_k1 n_k1 + sum_l G_kl x_l  (record of k_ia_solve_links: G[3][3], U[3][2], flag)
    static constexpr int K = 3, R = 2, J = 1, S = 3, STRIDE = 16, OK_AT = 15, PER_WAVE = 16;
    static constexpr bool FULL = true;
    __host__ __device__ static constexpr int gain(int s, int l) { return 3 * s + l; }
    __host__ __device__ static constexpr int mix(int s, int a) { return 9 + 2 * s + a; }
};
template <int KC, int RR> struct BdWalk {   // f6: est_s = d_s x_s + sum_a W_sa n_ka  (record of k_bd_solve_links: d[n], W[n][R], flag)
    static constexpr int K = KC, R = RR, J = RR, S = KC * RR, STRIDE = S * (RR + 1) + 1, OK_AT = S * (RR + 1), PER_WAVE = 8;
    static constexpr bool FULL = false;
    __host__ __device__ static constexpr int gain(int s, int) { return s; }
    __host__ __device__ static constexpr int mix(int s, int a) { return S + s * RR + a; }
};

constexpr int kWalkRunBytes = kBlocksPerRun * 16;       // 144: the DATA blocks under <= 128 consecutive positions

// ABL (MCLE_EXPERIMENTS builds, option f64_variant; WRONG results by construction -- the ablations behind the section tables):
// 1 = no symbol draws, 2 = no noise Philox blocks, 4 = no Box-Muller, 8 = no estimate arithmetic, 16 = no decisions
#ifndef MCLE_WALK_F64_WAVES
#define MCLE_WALK_F64_WAVES 3       // wavefronts per SIMD the registers are bounded for (A/B: profiles/r06/walk_f64_ab.log)
#endif
// T = double: the kernel described above (name kept in the profiles: k_link_walk<double, ...>).  T = float (round 6, late): the same
// packing, LDS records and label exchange for the complex64 walks; the decisions of a pass are taken together AFTER its estimates
// (twelve more registers are cheap in float) -- the slicer four at a time in the packed level domain, the certificates straight-line.
template <typename T, typename P, int DEC, int ABL = 0>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? MCLE_F32_WALK_WAVES : MCLE_WALK_F64_WAVES) void k_link_walk(ModemParams<T> mp, int n_symbols, T sigma, uint64_t seed,
                                                        uint64_t first, uint64_t count, const cx<T>* __restrict__ recs,
                                                        mcle_counters* counters, uint32_t* __restrict__ sym_out,
                                                        uint32_t* __restrict__ bit_out) {
    constexpr int S = P::S, R = P::R, J = P::J, K = P::K, PW = P::PER_WAVE, RUNS = 2 * S;
    extern __shared__ __attribute__((aligned(16))) unsigned char walk_smem[];    // [M] constellation, then the candidate grid (generic form)
    cx<T>* s_table = reinterpret_cast<cx<T>*>(walk_smem);
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(walk_smem + (((size_t)mp.M * sizeof(cx<T>) + 15) & ~(size_t)15));
    __shared__ double s_bm[sizeof(T) == 8 ? kBmLdsDoubles : 1];   // complex128: Box-Muller tables (bm_f64.hpp)
    // A workgroup is FOUR independent wavefronts that share the tables and ONE flush of the counters: with one wavefront per
    // workgroup the six global atomics of wg_flush -- 6 144 workgroups per launch on one cache line, ~9 ns each -- were a third of a
    // complex64 launch (profiles/r06/walk_grid_sweep.log: the time grew with the grid, not with the work).
    __shared__ cx<T> s_rec_all[4][PW * P::STRIDE];         // the chunk's records
    __shared__ uint4 s_sym_all[4][RUNS * kBlocksPerRun];   // the pass's DATA blocks, run by run
    __shared__ unsigned s_se_all[4][PW], s_be_all[4][PW];  // error counts of the chunk's realizations
    __shared__ WgTotals totals_all[4];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cx<T>* s_rec = s_rec_all[wv];
    uint4* s_sym = s_sym_all[wv];
    unsigned* s_se = s_se_all[wv];
    unsigned* s_be = s_be_all[wv];
    WgTotals& totals = totals_all[wv];
    if constexpr (sizeof(T) == 8) bm_tables_to_lds(s_bm, (int)threadIdx.x, (int)blockDim.x);
    load_table(mp, s_table);
    if constexpr (DEC == WDEC_GENERIC) load_grid(mp, s_grid);
    const int lane = threadIdx.x & 63;
    const uint32_t NS = (uint32_t)n_symbols, NP = NS >> 1, mask = (uint32_t)(mp.M - 1);
    if (lane == 0) wg_zero(totals);
    __syncthreads();
    // producer side of the symbol exchange: lane -> (run, block of the run); run = segment * S + stream
    const int p_run = lane / kBlocksPerRun, p_blk = lane - p_run * kBlocksPerRun;
    const uint64_t n_chunks = (count + PW - 1) / PW;
    for (uint64_t ch = (uint64_t)blockIdx.x * 4 + wv; ch < n_chunks; ch += (uint64_t)gridDim.x * 4) {
        const uint64_t rb = ch * PW;
        const int nr = (int)(count - rb < (uint64_t)PW ? count - rb : (uint64_t)PW);
        const uint32_t n_pairs = (uint32_t)nr * NP;
        {
            const cx<T>* src = recs + rb * P::STRIDE;
            for (int i = lane; i < nr * P::STRIDE; i += 64) s_rec[i] = src[i];
            if (lane < PW) {
                s_se[lane] = 0u;
                s_be[lane] = 0u;
            }
        }
        walk_wave_order();
        uint32_t rloc = 0, rem = (uint32_t)lane;           // lane's pair p = p0 + lane = rloc * NP + rem (NP >= 64)
        for (uint32_t p0 = 0; p0 < n_pairs; p0 += 64) {
            const bool valid = p0 + (uint32_t)lane < n_pairs;
            const uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)rloc);
            const bool two_seg = (uint32_t)__builtin_amdgcn_readlane((int)rloc, 63) != rlo;     // (wave-uniform)
            const uint32_t t_lo = 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
            const uint32_t seg = rloc - rlo, t = 2u * rem;
            int ta[S], tb[S];
            if constexpr (ABL & 1) {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    ta[k] = (int)((uint32_t)(lane + k) & mask);
                    tb[k] = (int)((uint32_t)(lane + 2 * k + 1) & mask);
                }
            } else {
#pragma unroll
                for (int u0 = 0; u0 < RUNS; u0 += kStreamsPerRound) {
                    const int u = u0 + p_run;
                    if (u0 >= S && !two_seg) break;          // runs of a second realization only: most passes have none
                    if (p_run < kStreamsPerRound && u < RUNS) {
                        const int sg = u >= S ? 1 : 0, k = u - sg * S;
                        const uint32_t base = (uint32_t)k * NS + (sg ? 0u : t_lo);
                        const Rng rg(seed, first + rb + rlo + (uint32_t)sg);
                        const Words4 b = rg.block(STREAM_DATA, (base >> 4) + (uint32_t)p_blk);
                        s_sym[u * kBlocksPerRun + p_blk] = make_uint4(b.w[0], b.w[1], b.w[2], b.w[3]);
                    }
                }
                walk_wave_order();
                const unsigned char* sym = reinterpret_cast<const unsigned char*>(s_sym);
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const uint32_t q = (uint32_t)k * NS + t;
                    const uint32_t base = (uint32_t)k * NS + (seg ? 0u : t_lo);
                    const uint32_t off = (seg * (uint32_t)S + (uint32_t)k) * (uint32_t)kWalkRunBytes + (q - (base & ~15u));
                    const uint32_t w = *reinterpret_cast<const unsigned short*>(sym + off);     // labels of columns t, t + 1
                    ta[k] = (int)(w & mask);
                    tb[k] = (int)((w >> 8) & mask);
                }
                walk_wave_order();
            }
            unsigned se = 0, be = 0;
            if (valid) {
                const Rng rng(seed, first + rb + rloc);
                const cx<T>* rc = s_rec + rloc * (uint32_t)P::STRIDE;
                cx<T> xa[P::FULL ? S : 1], xb[P::FULL ? S : 1];
                [[maybe_unused]] cx<T> e_all[sizeof(T) == 4 ? 2 * S : 1];        // complex64: the pass's estimates, decided together below
                [[maybe_unused]] int tx_all[sizeof(T) == 4 ? 2 * S : 1];
                if constexpr (P::FULL) {
#pragma unroll
                    for (int l = 0; l < S; ++l) {
                        xa[l] = s_table[ta[l]];
                        xb[l] = s_table[tb[l]];
                    }
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    cx<T> za[R], zb[R];
#pragma unroll
                    for (int a = 0; a < R; ++a) {
                        const uint32_t bi = ((uint32_t)(k * R + a) * NS + t) >> 1;
                        if constexpr ((ABL & 6) == 0) {
                            cn_pair_lds(rng, STREAM_NOISE, bi, sigma, za[a], zb[a], s_bm);
                        } else {
                            Words4 b;
                            if constexpr (ABL & 2) {      // four DIFFERENT cheap words: the two Box-Muller pairs must not fold into one
                                b.w[0] = bi * 2654435769u;
                                b.w[1] = b.w[0] ^ 0x9E3779B9u;
                                b.w[2] = b.w[0] + 0x7F4A7C15u;
                                b.w[3] = b.w[2] ^ 0x85EBCA6Bu;
                            } else {
                                b = rng.block(STREAM_NOISE, bi);
                            }
                            if constexpr (ABL & 4) {
                                za[a] = mk<T>((T)(int)b.w[0] * (T)1e-10, (T)(int)b.w[1] * (T)1e-10);
                                zb[a] = mk<T>((T)(int)b.w[2] * (T)1e-10, (T)(int)b.w[3] * (T)1e-10);
                            } else if constexpr (sizeof(T) == 8) {
                                za[a] = cn_from_words_lds(b.w[0], b.w[1], sigma, s_bm);
                                zb[a] = cn_from_words_lds(b.w[2], b.w[3], sigma, s_bm);
                            } else {
                                za[a] = cn_from_words(b.w[0], b.w[1], sigma);
                                zb[a] = cn_from_words(b.w[2], b.w[3], sigma);
                            }
                        }
                    }
                    cx<T> e[2 * J];
                    int tx[2 * J];
#pragma unroll
                    for (int jj = 0; jj < J; ++jj) {
                        const int s = k * J + jj;
                        cx<T> ea, eb;
                        if constexpr (ABL & 8) {          // every noise sample and the symbol stay alive, the multiply-adds go
                            ea = cadd(cadd(za[0], za[R - 1]), P::FULL ? xa[s] : s_table[ta[s]]);
                            eb = cadd(cadd(zb[0], zb[R - 1]), P::FULL ? xb[s] : s_table[tb[s]]);
                        } else if constexpr (P::FULL) {
                            {
                                const cx<T> c = rc[P::mix(s, 0)];
                                ea = cmul(c, za[0]);
                                eb = cmul(c, zb[0]);
                            }
#pragma unroll
                            for (int a = 1; a < R; ++a) {
                                const cx<T> c = rc[P::mix(s, a)];
                                ea = cfma4(c, za[a], ea);
                                eb = cfma4(c, zb[a], eb);
                            }
#pragma unroll
                            for (int l = 0; l < S; ++l) {
                                const cx<T> c = rc[P::gain(s, l)];
                                ea = cfma4(c, xa[l], ea);
                                eb = cfma4(c, xb[l], eb);
                            }
                        } else {
                            {
                                const cx<T> c = rc[P::gain(s, s)];
                                ea = cmul(c, s_table[ta[s]]);
                                eb = cmul(c, s_table[tb[s]]);
                            }
#pragma unroll
                            for (int a = 0; a < R; ++a) {
                                const cx<T> c = rc[P::mix(s, a)];
                                ea = cfma4(c, za[a], ea);
                                eb = cfma4(c, zb[a], eb);
                            }
                        }
                        e[2 * jj] = ea;
                        e[2 * jj + 1] = eb;
                        tx[2 * jj] = ta[s];
                        tx[2 * jj + 1] = tb[s];
                    }
                    if constexpr (ABL & 16) {
#pragma unroll
                        for (int j = 0; j < 2 * J; j += 2)      // both components of both estimates stay alive
                            se += (unsigned)(e[j].x + e[j].y > e[j + 1].x + e[j + 1].y) + (unsigned)(tx[j] > tx[j + 1]);
                    } else if constexpr (sizeof(T) == 8) {
                        walk_decide<T, DEC, 2 * J>(mp, s_table, s_grid, e, tx, se, be);
                    } else {
#pragma unroll
                        for (int j = 0; j < 2 * J; ++j) {
                            e_all[2 * J * k + j] = e[j];
                            tx_all[2 * J * k + j] = tx[j];
                        }
                    }
                    if constexpr (sizeof(T) == 8)
                        __builtin_amdgcn_sched_barrier(0);  // user by user: the next user's draws do not start under this one's tail
                }
                if constexpr (sizeof(T) == 4 && !(ABL & 16)) walk_decide<T, DEC, 2 * S>(mp, s_table, s_grid, e_all, tx_all, se, be);
            }
            // the pass's counts to the (at most two) realizations it covers: per lane se <= 2 S, be <= 16 S, 64 lanes -- 16 bits each
            {
                const uint32_t w = se | (be << 16);
                const uint32_t tot = walk_wave_total(w);
                uint32_t hi = 0;
                if (two_seg) hi = walk_wave_total(seg ? w : 0u);
                if (lane == 0) {                        // (LDS atomics without return: nothing waits for them)
                    const uint32_t lo = tot - hi;
                    atomicAdd(&s_se[rlo], lo & 0xFFFFu);
                    atomicAdd(&s_be[rlo], lo >> 16);
                    if (hi) {
                        atomicAdd(&s_se[rlo + 1], hi & 0xFFFFu);
                        atomicAdd(&s_be[rlo + 1], hi >> 16);
                    }
                }
            }
            rem += 64u;
            if (rem >= NP) {
                rem -= NP;
                ++rloc;
            }
        }
        walk_wave_order();
        if (lane == 0)
            for (int i = 0; i < nr; ++i)
                wg_account(totals, s_se[i], s_be[i], s_rec[i * P::STRIDE + P::OK_AT].x == 0.0, rb + (uint64_t)i, sym_out, bit_out);
        walk_wave_order();
    }
    wg_flush_waves<4>(totals_all, counters, (unsigned long long)S * NS, (unsigned long long)S * NS * (unsigned long long)mp.bits);
}

// host: does this request fit the kernel above?  (an even number of columns, at least 128 of them: a pass then covers at most
// two realizations; label bytes: M <= 256)
inline bool link_walk_f64_fits(int n_symbols) { return (n_symbols & 1) == 0 && n_symbols >= 128; }

template <typename T, typename P, int ABL = 0>
inline void launch_link_walk(mcle_ctx* ctx, const ModemParams<T>& mp_in, int n_symbols, double noise_var, uint64_t seed,
                             uint64_t first, uint64_t count, const cx<T>* recs, mcle_counters* d_counters, uint32_t* d_sym,
                             uint32_t* d_bit) {
    ModemParams<T> mp = mp_in;
    const int dec = walk_dec_kind(ctx, mp);
    if (dec != WDEC_GENERIC) mp.grid.G = 0;
    const size_t lds = (((size_t)mp.M * sizeof(cx<T>) + 15) & ~(size_t)15) + (size_t)mp.grid.G * mp.grid.G * sizeof(unsigned long long);
    const uint64_t chunks = (count + P::PER_WAVE - 1) / P::PER_WAVE;
    const uint64_t cap = (uint64_t)ctx->n_cu * (sizeof(T) == 4 ? MCLE_F32_WALK_WAVES : MCLE_WALK_F64_WAVES);     // workgroups of four wavefronts
    const unsigned grid = (unsigned)oversubscribed_grid(ctx, cap, (chunks + 3) / 4, 2);
    const T sigma = (T)sqrt(noise_var);
    auto kern = k_link_walk<T, P, WDEC_GENERIC, ABL>;
    switch (dec) {
        case WDEC_SLICER: kern = k_link_walk<T, P, WDEC_SLICER, ABL>; break;
        case WDEC_QAM_CERT: kern = k_link_walk<T, P, WDEC_QAM_CERT, ABL>; break;
        case WDEC_QUAD_CERT: kern = k_link_walk<T, P, WDEC_QUAD_CERT, ABL>; break;
        case WDEC_AXIS4_CERT: kern = k_link_walk<T, P, WDEC_AXIS4_CERT, ABL>; break;
        default: break;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, mp, n_symbols, sigma, seed, first, count, recs, d_counters,
                       d_sym, d_bit);
}

}  // namespace mcle

namespace mcle {
// (round-6 name of the complex128 launcher, kept for the callers written against it)
template <typename P, int ABL = 0>
inline void launch_link_walk_f64(mcle_ctx* ctx, const ModemParams<double>& mp, int n_symbols, double noise_var, uint64_t seed, uint64_t first,
                                 uint64_t count, const double2* recs, mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    launch_link_walk<double, P, ABL>(ctx, mp, n_symbols, noise_var, seed, first, count, recs, d_counters, d_sym, d_bit);
}
}  // namespace mcle
