// mimo_tdl.hpp -- what the frequency-selective MIMO-OFDM link kernels share: the parameter block and
// the kernel that turns a symbol's Jakes rays into tap polynomials (pipeline_mimo_tdl.hip: the workgroup-cooperative kernel of
// rounds 1-4; mimo_tdl_wave.hpp: one receive antenna per wavefront, round 5).
#pragma once
#include "bm_f64.hpp"
#include "common.hpp"
#include "jakes.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kMaxOrder = 12;
#ifndef FFT_FRESH
#define FFT_FRESH true
#endif

struct MimoTdlParams {
    int cp, num_used, n_ofdm_sym, mmse;
    int n_taps, L, K, dmax;
    int x_elems;                     // complex elements of the sample buffer (>= NA*N; also holds the ray scratch)
    double noise_var, Fd, Ts, dt;
    double tap_amp[MCLE_MAX_TAPS];   // sqrt(p_s / L)
    int tap_delay[MCLE_MAX_TAPS];
    double mom[kMaxOrder + 1];       // mean over the symbol's N+cp samples of x^m, x = j - (N+cp-1)/2
    // the wavefront kernels (mimo_tdl_wave.hpp): tap s expanded around the symbol centre MINUS its delay, so that the channel stage
    // evaluates every tap at the output sample's abscissa; mean over the symbol's samples of (j + d_s - (N+cp-1)/2)^m
    double mom_tap[8][kMaxOrder + 1];
    // the decode of the wavefront kernels where a lane holds the bins f0 and f0 + N / 2: w^((f0 + N / 2) d) = (-1)^d w^(f0 d), so
    // the taps are summed by the parity of their delay and the two bins are the sum and the difference.  Class POSITIONS: the
    // even-delay taps at positions 0 .. cls_ne - 1, the odd-delay ones at 7, 6, .. 8 - cls_no (both loops then index registers
    // statically); cls_code[p] = tap << 16 | delay, -1 where the position is empty
    int cls_code[8];
    int cls_ne, cls_no;
};

// The fading of one OFDM symbol in its own launch (round 3, as k_tdl_symbol_polys did for config 3): one thread per
// (realization, symbol, fading process p = (tap s, rx r, tx a)) folds the process's L rays -- phasor at the symbol centre and
// phase advance per sample, the f64 phase arithmetic of fading_generators.py:427-493 -- into the K + 1 polynomial
// coefficients of g_p(x) around the centre and their mean over the symbol (the equaliser's tap).  Inside the link kernel this
// ran on 2.5 rounds of 256 threads between three workgroup barriers per symbol: 10 % of its time for 3 % of its arithmetic.
// Record per (realization, symbol): [PS][K + 1] coefficients, then [PS] means; same operations in the same order as before.
// WAVE (round 5, mimo_tdl_wave.hpp: one receive antenna per wavefront): the coefficients in the order a wavefront parks them
// across its lanes -- receive antenna r, register q = m / 2, lane 2 (s Nt + a) + m % 2 (P1 = Nr Nt, NT = Nt): record =
// [Nr][NQ][LW] coefficients, NQ = (K + 2) / 2 registers of LW = 2 S Nt lanes, then the [PS] means; the same values.
__host__ __device__ __forceinline__ int mimo_tdl_nq(int K) { return (K + 2) / 2; }
__host__ __device__ __forceinline__ size_t mimo_tdl_wave_rec(int S, int NT, int NR, int K) {
    return (size_t)NR * mimo_tdl_nq(K) * (2 * S * NT) + (size_t)S * NR * NT;
}
// KT > 0 (last day of round 6): the polynomial order at compile time -- the run-time form walks a ladder of thirteen uniform
// branches per ray for its `m <= K` guards; I32: count x processes below 2^32, the thread's (realization, process) split in 32-bit
// arithmetic (the 64-bit division is ~100 instructions).  Same operations on the same values in the same order.
template <typename T, bool WAVE = false, int KT = 0, bool I32 = false>
__global__ __launch_bounds__(256) void k_mimo_tdl_symbol_polys(MimoTdlParams pp, int PS, int P1, int W, uint64_t seed,
                                                               uint64_t first, uint64_t count, cx<T>* __restrict__ recs,
                                                               int NT = 0) {
    const int L = pp.L, K = KT > 0 ? KT : pp.K;
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * PS;
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count * per_real) return;
    uint64_t rl;
    int rem;
    if constexpr (I32) {
        const uint32_t q32 = (uint32_t)q, pr32 = (uint32_t)per_real, r32 = q32 / pr32;
        rl = r32;
        rem = (int)(q32 - r32 * pr32);
    } else {
        rl = q / per_real;
        rem = (int)(q - rl * per_real);
    }
    const int os = rem / PS, p = rem - os * PS;
    const double two_pi = 6.283185307179586476925286766559;
    const double xc = 0.5 * (double)(W - 1);
    // WAVE: the polynomial of tap s in the OUTPUT sample's abscissa x' = x + d_s -- the same rays about the centre minus d_s samples
    const int tap = p / P1;
    const double tc = pp.Ts + pp.dt * ((double)((uint64_t)os * W) + (WAVE ? xc - (double)pp.tap_delay[tap] : xc));
    const Rng rng(seed, first + rl);
    // (the largest |Doppler phase| any sample of the run can see, in turns)
    [[maybe_unused]] const bool small_phase = pp.Fd * (pp.Ts + pp.dt * ((double)(pp.n_ofdm_sym + 1) * W + (double)pp.dmax)) < 0.25;
    T ar[kMaxOrder + 1], ai[kMaxOrder + 1];
#pragma unroll
    for (int m = 0; m <= kMaxOrder; ++m) ar[m] = ai[m] = 0;
    // PHASE draws: uniform i of the stream is word i % 4 of block i / 4, and the four processes p = 4 g .. 4 g + 3 of a quad of
    // lanes read the four words of the SAME blocks (PS a multiple of 4).  Each lane of the quad computes ONE of the four blocks
    // a pair of rays needs -- (phi, psi) of rays l and l + 1 -- and the words travel by DPP quad broadcasts: a quarter of the
    // Philox work of one block per uniform (the ledger is untouched: the same words reach the same processes).
    const bool quad = (PS & 3) == 0;
    const int ql = (int)(threadIdx.x & 3);                                  // = p % 4 (256 and every record boundary are multiples of 4)
    double u_phi_next = 0.0, u_psi_next = 0.0;
    for (int l = 0; l < L; ++l) {
        const uint64_t rq = (uint64_t)l * PS + p;                          // PHASE-stream index of phi
        double u_phi, u_psi;
        if (quad && (l & 1) == 0 && l + 1 < L) {
            const uint64_t mine = (uint64_t)(l + (ql >> 1)) * PS + p + ((ql & 1) ? (uint64_t)L * PS : 0ull);
            const Words4 b = rng.block(STREAM_PHASE, (uint32_t)(mine >> 2));
            uint32_t got[4];                                               // word ql of the block lane j of the quad computed
#define MCLE_QUAD_WORD(J)                                                                                                     \
    {                                                                                                                         \
        uint32_t v = 0;                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                       \
            const uint32_t t = (uint32_t)__builtin_amdgcn_mov_dpp((int)b.w[k], (J) * 0x55, 0xF, 0xF, true); /* quad_perm [J, J, J, J] */ \
            v = ql == k ? t : v;                                                                                              \
        }                                                                                                                     \
        got[J] = v;                                                                                                           \
    }
            MCLE_QUAD_WORD(0) MCLE_QUAD_WORD(1) MCLE_QUAD_WORD(2) MCLE_QUAD_WORD(3)
#undef MCLE_QUAD_WORD
            u_phi = (double)got[0] * 0x1p-32;
            u_psi = (double)got[1] * 0x1p-32;
            u_phi_next = (double)got[2] * 0x1p-32;
            u_psi_next = (double)got[3] * 0x1p-32;
        } else if (quad && (l & 1) == 1) {
            u_phi = u_phi_next;
            u_psi = u_psi_next;
        } else {
            u_psi = uniform_at(rng, STREAM_PHASE, (uint64_t)L * PS + rq);
            u_phi = uniform_at(rng, STREAM_PHASE, rq);
        }
        const double psi_t = u_psi;
        T er, ei, th;
        bool done = false;
        if constexpr (sizeof(T) == 4) {
            // complex64 with the Doppler phase of the whole run below a quarter turn (config f1: 1e-3): the ray's frequency from
            // v_cos_f32 -- its 1.5e-6 of absolute error is then < 4e-7 turns of phase, below the v_sin / v_cos of the phasor itself --
            // instead of the f64 cospi (45 instructions per ray, a fifth of this kernel)
            if (small_phase) {
                const float wf = (float)pp.Fd * __builtin_amdgcn_cosf((float)u_phi);      // (v_cos_f32 takes turns)
                const float fr = __builtin_amdgcn_fractf(fmaf(wf, (float)tc, (float)psi_t));
                er = __builtin_amdgcn_cosf(fr);
                ei = __builtin_amdgcn_sinf(fr);
                th = (float)(two_pi * pp.dt) * wf;
                done = true;
            }
        }
        if (!done) {
            const double w = pp.Fd * cospi(2.0 * u_phi);                   // Hz; cos(phi), phi = 2 pi u
            const double ph = fma(w, tc, psi_t);                           // turns
            const double fr = __builtin_amdgcn_fract(ph);
            if constexpr (sizeof(T) == 8) {
                double sn, cs;
                bm_sincos_rad(two_pi * fr, cs, sn);                        // polynomial form (bm_f64.hpp): a third of the library's instructions
                er = cs;
                ei = sn;
            } else {
                er = __builtin_amdgcn_cosf((float)fr);
                ei = __builtin_amdgcn_sinf((float)fr);
            }
            th = (T)(two_pi * w * pp.dt);                                  // rad per sample
        }
#pragma unroll
        for (int m = 0; m <= (KT > 0 ? KT : kMaxOrder); ++m)
            if (KT > 0 || m <= K) {
                T pw = 1;                                                  // 1 / m! ...
                for (int i = 2; i <= m; ++i) pw /= (T)i;
                for (int i = 0; i < m; ++i) pw *= th;                      // ... x theta^m
                ar[m] += er * pw;
                ai[m] += ei * pw;
            }
    }
    const T amp = (T)pp.tap_amp[p / P1];
    const int S = PS / P1;
    [[maybe_unused]] const int NR = WAVE ? P1 / NT : 0, NQ = mimo_tdl_nq(K), LW = WAVE ? 2 * S * NT : 0;
    cx<T>* rec = recs + (rl * pp.n_ofdm_sym + os) * (WAVE ? (uint64_t)mimo_tdl_wave_rec(S, NT, NR, K) : (uint64_t)PS * (K + 2));
    T mr = 0, mi = 0;
#pragma unroll
    for (int m = 0; m <= (KT > 0 ? KT : kMaxOrder); ++m)
        if (KT > 0 || m <= K) {
            T cr, ci;                                                      // times j^m
            switch (m & 3) {
                case 0: cr = ar[m]; ci = ai[m]; break;
                case 1: cr = -ai[m]; ci = ar[m]; break;
                case 2: cr = -ar[m]; ci = -ai[m]; break;
                default: cr = ai[m]; ci = -ar[m]; break;
            }
            const cx<T> c = mk<T>(amp * cr, amp * ci);
            if constexpr (WAVE) {
                const int s = p / P1, ra = p - s * P1, r = ra / NT, a = ra - r * NT;
                rec[(r * NQ + (m >> 1)) * LW + 2 * (s * NT + a) + (m & 1)] = c;
            } else {
                rec[p * (K + 1) + m] = c;
            }
            const T mo = (T)(WAVE ? pp.mom_tap[tap & 7][m] : pp.mom[m]);
            mr += c.x * mo;
            mi += c.y * mo;
        }
    rec[(WAVE ? NR * NQ * LW : PS * (K + 1)) + p] = mk<T>(mr, mi);
}
// host: the records of n realizations starting at `first` (one launch on `stream`): the compile-time orders of the benchmark's
// Doppler per arithmetic (2, 5) and the one between, 32-bit index arithmetic where the thread count allows
template <typename T, bool WAVE>
inline void launch_mimo_tdl_symbol_polys(hipStream_t stream, const MimoTdlParams& pp, int PS, int P1, int W, uint64_t seed, uint64_t first,
                                         uint64_t n, cx<T>* recs, int NT) {
    const uint64_t threads = n * (uint64_t)pp.n_ofdm_sym * PS;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    const bool i32 = threads < (1ull << 32);
#define MCLE_POLYS_K(KT_)                                                                                                       \
    if (i32) hipLaunchKernelGGL((k_mimo_tdl_symbol_polys<T, WAVE, KT_, true>), grid, block, 0, stream, pp, PS, P1, W, seed, first, n, recs, NT); \
    else hipLaunchKernelGGL((k_mimo_tdl_symbol_polys<T, WAVE, KT_, false>), grid, block, 0, stream, pp, PS, P1, W, seed, first, n, recs, NT);
    switch (pp.K) {
        case 2: MCLE_POLYS_K(2) break;
        case 3: MCLE_POLYS_K(3) break;
        case 5: MCLE_POLYS_K(5) break;
        default: MCLE_POLYS_K(0) break;
    }
#undef MCLE_POLYS_K
}

}  // namespace mcle
