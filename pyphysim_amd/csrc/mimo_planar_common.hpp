// mimo_planar_common.hpp -- what the planar config-4 kernels (pipeline_mimo_planar.hip) and the quarter-wave kernel
// (pipeline_mimo_qw.hip) share: the parameter block, the per-realization record (channel + receive filter) and its kernel, the
// lane-swap and Box-Muller helpers.  Moved out of pipeline_mimo_planar.hip in round 6, unchanged.
#pragma once
#include <type_traits>
#include <utility>

#include "fft.hpp"
#include "fft_r16.hpp"
#include "mimo.hpp"
#include "modem.hpp"
#include "philox.hpp"
#include "totals.hpp"
#include "pipe_common.hpp"

namespace mcle {

struct MimoParams {
    int cp, num_used, n_ofdm_sym;
    int mmse;
    double noise_var;
};

template <int NT, int NR> constexpr int d64_rec() { return 2 * NT * NR + 1; }     // H, G x FFT scale, skip flag


// complex64: the channel is drawn in float (the draw ledger of the complex64 kernels), the filter computed in double and rounded
template <typename T, int N, int NT, int NR>
__global__ __launch_bounds__(64) void k_mimo_filters_planar(MimoParams pp, uint64_t seed, uint64_t first, uint64_t count,
                                                            cx<T>* __restrict__ recs) {
    constexpr int kRec = d64_rec<NT, NR>();
    const uint64_t rl = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (rl >= count) return;
    const double rx_scale = sqrt((double)(pp.num_used + pp.cp)) / (double)N;
    const Rng rng(seed, first + rl);
    cx<T>* rec = recs + rl * kRec;
    double2 H[NR][NT], G[NT][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const cx<T> h = cn_sample<T>(rng, STREAM_CHAN, (uint64_t)(r * NT + a), (T)1);
            rec[r * NT + a] = h;
            H[r][a] = mk<double>((double)h.x, (double)h.y);
        }
    const bool ok = blast_filter<NT, NR>(H, pp.mmse ? pp.noise_var : 0.0, G);
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < NR; ++r) rec[NT * NR + a * NR + r] = mk<T>((T)(G[a][r].x * rx_scale), (T)(G[a][r].y * rx_scale));
    rec[2 * NT * NR] = mk<T>(ok ? (T)0 : (T)1, (T)0);
}

// lanes l and l ^ 32 exchange: (a of the lower half, b of the upper half) stay, the other two cross over --
// x = {lower: own a, upper: the partner's b}, y = {lower: the partner's a, upper: own b}  (v_permlane32_swap_b32)
__device__ __forceinline__ void swap32_pair(double a, double b, double& x, double& y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap32_pair(float a, float b, float& x, float& y) {
    const auto v = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    x = __uint_as_float(v[0]);
    y = __uint_as_float(v[1]);
}

// one CN(0, sigma^2) sample from two Philox words: complex128 through the LDS Box-Muller tables, complex64 by the hardware
// transcendentals (the complex64 kernels' draw)
__device__ __forceinline__ double2 cn_words(uint32_t x0, uint32_t x1, double sigma, const double* s_bm) {
    return cn_from_words_lds(x0, x1, sigma, s_bm);
}
__device__ __forceinline__ float2 cn_words(uint32_t x0, uint32_t x1, float sigma, const double*) { return cn_from_words(x0, x1, sigma); }

}  // namespace mcle
