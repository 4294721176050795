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
// SH (round 6): the PHASE blocks of a workgroup's G realizations are evaluated ONCE into LDS (2 L S / 4 blocks per realization: 20 for
// config 3) and every thread reads its 2 L uniforms from there -- with one Philox call per uniform (uniform_at) a thread evaluated 2 L
// = 16 blocks and used one word of each: 80 block evaluations per realization for 20 blocks.  Same words, same operations after them.
template <typename T, bool SH = false>
__global__ __launch_bounds__(256) void k_tdl_symbol_polys(SisoTdlParams pp, int W, uint64_t seed, uint64_t first, uint64_t count,
                                                          cx<T>* __restrict__ recs, int G = 0, int NB = 0) {
    const int S = pp.n_taps, L = pp.L, K = pp.K;
    const int per = pp.n_ofdm_sym * S;
    extern __shared__ uint32_t s_phase_words[];                             // SH: [G][4 NB]
    uint64_t rl;
    int rem;
    [[maybe_unused]] const uint32_t* my_words = nullptr;
    if constexpr (SH) {
        const uint64_t r0 = (uint64_t)blockIdx.x * G;
        for (int i = (int)threadIdx.x; i < G * NB; i += 256) {
            const int g = i / NB, b = i - g * NB;
            if (r0 + g < count) {
                const Words4 wq = Rng(seed, first + r0 + g).block(STREAM_PHASE, (uint32_t)b);
                *reinterpret_cast<uint4*>(s_phase_words + 4 * i) = make_uint4(wq.w[0], wq.w[1], wq.w[2], wq.w[3]);
            }
        }
        __syncthreads();
        const int g = (int)threadIdx.x / per;
        rem = (int)threadIdx.x - g * per;
        rl = r0 + g;
        if (g >= G || rl >= count) return;
        my_words = s_phase_words + 4 * NB * g;
    } else {
        const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (q >= count * (uint64_t)per) return;
        rl = q / (uint64_t)per;
        rem = (int)(q - rl * (uint64_t)per);
    }
    const int os = rem / S, s = rem - os * S;
    const double xc = 0.5 * (double)(W - 1);
    const double two_pi = 6.283185307179586476925286766559;
    const double tc = pp.Ts + pp.dt * ((double)((uint64_t)os * W) + xc);
    const Rng rng(seed, first + rl);
    [[maybe_unused]] const bool small_phase = pp.Fd * (pp.Ts + pp.dt * ((double)(pp.n_ofdm_sym + 1) * W + 256.0)) < 0.25;   // turns, any sample of the run
    T ar[kTdlMaxK + 1], ai[kTdlMaxK + 1];
#pragma unroll
    for (int m = 0; m <= kTdlMaxK; ++m) ar[m] = ai[m] = 0;
    for (int l = 0; l < L; ++l) {
        const int rq = l * S + s;                                         // PHASE-stream index of phi
        double psi_t, u_phi;
        if constexpr (SH) {
            psi_t = (double)my_words[L * S + rq] * 0x1p-32;
            u_phi = (double)my_words[rq] * 0x1p-32;
        } else {
            psi_t = uniform_at(rng, STREAM_PHASE, (uint64_t)L * S + rq);
            u_phi = uniform_at(rng, STREAM_PHASE, (uint64_t)rq);
        }
        T er, ei, th;
        bool done = false;
        if constexpr (sizeof(T) == 4) {
            // complex64 with the Doppler phase of the whole run below a quarter turn: the ray's frequency from v_cos_f32 (its 1.5e-6 of
            // absolute error is then < 4e-7 turns of phase) instead of the f64 cospi -- as in k_mimo_tdl_symbol_polys (mimo_tdl.hpp)
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
            const double wd = pp.Fd * cospi(2.0 * u_phi);                 // Hz
            const double ph = fma(wd, tc, psi_t);                         // turns
            const double fr = __builtin_amdgcn_fract(ph);
            if constexpr (sizeof(T) == 8) {
                double sn, cs;
                sincos(two_pi * fr, &sn, &cs);
                er = cs;
                ei = sn;
            } else {
                er = __builtin_amdgcn_cosf((float)fr);
                ei = __builtin_amdgcn_sinf((float)fr);
            }
            th = (T)(two_pi * wd * pp.dt);                                // rad per sample
        }
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

// host: the records of n realizations starting at `first` (one launch on `stream`)
template <typename T>
inline void launch_tdl_symbol_polys(hipStream_t stream, const SisoTdlParams& pp, int W, uint64_t seed, uint64_t first, uint64_t n,
                                    cx<T>* recs) {
    const int per = pp.n_ofdm_sym * pp.n_taps;
    const int NB = (2 * pp.L * pp.n_taps + 3) / 4;
    const int G = per <= 256 ? 256 / per : 0;
#ifndef MCLE_TDL_POLYS_SH
#define MCLE_TDL_POLYS_SH 1
#endif
    if (MCLE_TDL_POLYS_SH && G >= 1 && (size_t)G * NB * 16 <= (size_t)32 * 1024) {
        hipLaunchKernelGGL((k_tdl_symbol_polys<T, true>), dim3((unsigned)((n + G - 1) / G)), dim3(256), (size_t)G * NB * 16, stream, pp, W, seed,
                           first, n, recs, G, NB);
    } else {
        const uint64_t threads = n * (uint64_t)per;
        hipLaunchKernelGGL((k_tdl_symbol_polys<T, false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, pp, W, seed, first, n,
                           recs, 0, 0);
    }
}


}  // namespace mcle
