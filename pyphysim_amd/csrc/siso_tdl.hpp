// siso_tdl.hpp -- what the config-3 link kernels share: the parameter block and the kernel that turns a symbol's Jakes rays into tap
// polynomials (pipeline_siso_tdl.hip: the batched kernels; siso_tdl_wave.hpp: one realization per wavefront).
#pragma once
#include "common.hpp"
#include "jakes.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kSisoMaxOrder = 12;

struct SisoTdlParams {
    int cp, num_used, n_ofdm_sym;
    int n_taps, L, K, dmax;
    int x_elems;                     // complex elements of the sample buffer (>= NB*N; also holds the ray scratch)
    double noise_var, Fd, Ts, dt;
    double tap_amp[MCLE_MAX_TAPS];   // sqrt(p_s / L)
    int tap_delay[MCLE_MAX_TAPS];
    double mom[kSisoMaxOrder + 1];   // mean over the symbol's N+cp samples of x^m, x = j - (N+cp-1)/2
};

// The fading of a symbol, one thread per (realization, OFDM symbol, tap), in a launch of its own (round 3): the L rays of the
// tap (f64 phase at the symbol centre, PHASE stream), their fold into the tap polynomial c_m = amp sum_l e_l (j theta_l)^m / m!
// and the per-symbol tap mean sum_m c_m mom_m.  Inside the link kernels this work ran on a fraction of the 256 threads between
// workgroup barriers the other wavefronts waited at, and its registers were allocated for the whole kernel.
// Record of (realization, symbol): coef [S][K + 1], mean [S] -- S (K + 2) complex values (160 B for config 3 in complex64).
// W = samples per OFDM symbol (FFT + CP).  Same operations in the same order as the in-kernel form it replaces.
constexpr int kTdlMaxK = kSisoMaxOrder;
template <typename T>
__global__ __launch_bounds__(256) void k_tdl_symbol_polys(SisoTdlParams pp, int W, uint64_t seed, uint64_t first, uint64_t count,
                                                          cx<T>* __restrict__ recs) {
    const int S = pp.n_taps, L = pp.L, K = pp.K;
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_real = (uint64_t)pp.n_ofdm_sym * S;
    if (q >= count * per_real) return;
    const uint64_t rl = q / per_real;
    const int rem = (int)(q - rl * per_real), os = rem / S, s = rem - os * S;
    const double xc = 0.5 * (double)(W - 1);
    const double two_pi = 6.283185307179586476925286766559;
    const double tc = pp.Ts + pp.dt * ((double)((uint64_t)os * W) + xc);
    const Rng rng(seed, first + rl);
    T ar[kTdlMaxK + 1], ai[kTdlMaxK + 1];
#pragma unroll
    for (int m = 0; m <= kTdlMaxK; ++m) ar[m] = ai[m] = 0;
    for (int l = 0; l < L; ++l) {
        const int rq = l * S + s;                                         // PHASE-stream index of phi
        const double psi_t = uniform_at(rng, STREAM_PHASE, (uint64_t)L * S + rq);
        const double wd = pp.Fd * cospi(2.0 * uniform_at(rng, STREAM_PHASE, (uint64_t)rq));   // Hz
        const double ph = fma(wd, tc, psi_t);                             // turns
        const double fr = __builtin_amdgcn_fract(ph);
        T er, ei;
        if constexpr (sizeof(T) == 8) {
            double sn, cs;
            sincos(two_pi * fr, &sn, &cs);
            er = cs;
            ei = sn;
        } else {
            er = __builtin_amdgcn_cosf((float)fr);
            ei = __builtin_amdgcn_sinf((float)fr);
        }
        const T th = (T)(two_pi * wd * pp.dt);                            // rad per sample
#pragma unroll
        for (int m = 0; m <= kTdlMaxK; ++m)
            if (m <= K) {
                T pw = 1;                                                 // 1 / m! ...
                for (int i = 2; i <= m; ++i) pw /= (T)i;
                for (int i = 0; i < m; ++i) pw *= th;                     // ... x theta^m, in the order of the fused kernel
                ar[m] += er * pw;
                ai[m] += ei * pw;
            }
    }
    const T amp = (T)pp.tap_amp[s];
    cx<T>* rec = recs + (rl * pp.n_ofdm_sym + os) * (uint64_t)(S * (K + 2));
    T mr = 0, mi = 0;
#pragma unroll
    for (int m = 0; m <= kTdlMaxK; ++m)
        if (m <= K) {
            T cr, ci;                                                     // times j^m
            switch (m & 3) {
                case 0: cr = ar[m]; ci = ai[m]; break;
                case 1: cr = -ai[m]; ci = ar[m]; break;
                case 2: cr = -ar[m]; ci = -ai[m]; break;
                default: cr = ai[m]; ci = -ar[m]; break;
            }
            const cx<T> c = mk<T>(amp * cr, amp * ci);
            rec[s * (K + 1) + m] = c;
            mr += c.x * (T)pp.mom[m];
            mi += c.y * (T)pp.mom[m];
        }
    rec[S * (K + 1) + s] = mk<T>(mr, mi);
}


}  // namespace mcle
