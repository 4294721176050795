// modem.hpp -- device-side constellation helpers shared by operator kernels and pipelines.
// Stands in for Modulator.modulate / demodulate (reference modulators/fundamental.py:175-248)
// and util/misc.py:449-566 (count_bits / count_bit_errors).
#pragma once
#include "common.hpp"

namespace mcle {

template <typename T> struct ModemParams {
    const cx<T>* g_table;  // global constellation table [M]
    int M;
    int bits;        // log2(M)
    int method;      // MCLE_DEMOD_*
    T qam_scale;     // sqrt(2(M-1)/3)
    int qam_L;       // sqrt(M)
    int half_bits;   // bits/2
};

// exhaustive minimum distance over a table held in LDS; strict '<' keeps the FIRST minimum,
// which is numpy.argmin's tie rule (fundamental.py:245).  The reference compares |c - r|; we
// compare |c - r|^2 (same ordering away from rounding-level ties).
template <typename T>
__device__ __forceinline__ int demod_mindist(const cx<T>* __restrict__ s_table, int M, cx<T> r) {
    T best = (r.x - s_table[0].x) * (r.x - s_table[0].x) + (r.y - s_table[0].y) * (r.y - s_table[0].y);
    int idx = 0;
#pragma unroll 8
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

// K symbols against the same table in one sweep (one LDS read per candidate serves all K).  For f32
// the ordering metric is |c|^2/2 - Re(r conj(c)) (two FMAs per candidate; argmin-equivalent to
// |c - r|^2 since |r|^2 is common), read from a float4 table {c.re, c.im, |c|^2/2, 0}; f64 keeps the
// literal |c - r|^2 of the parity path.  First minimum wins, as in demod_mindist.
template <int K>
__device__ __forceinline__ void demod_mindist_multi(const float4* __restrict__ s_tab4, int M, const float2 (&r)[K],
                                                    int (&idx)[K]) {
    float best[K];
    {
        const float4 c = s_tab4[0];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            best[k] = fmaf(-r[k].y, c.y, fmaf(-r[k].x, c.x, c.z));
            idx[k] = 0;
        }
    }
#pragma unroll 4
    for (int m = 1; m < M; ++m) {
        const float4 c = s_tab4[m];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = fmaf(-r[k].y, c.y, fmaf(-r[k].x, c.x, c.z));
            if (d < best[k]) {
                best[k] = d;
                idx[k] = m;
            }
        }
    }
}
template <int K>
__device__ __forceinline__ void demod_mindist_multi(const double2* __restrict__ s_table, int M, const double2 (&r)[K],
                                                    int (&idx)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) idx[k] = demod_mindist<double>(s_table, M, r[k]);
}

__device__ __forceinline__ int gray2binary8(int g) {
    g ^= g >> 4;
    g ^= g >> 2;
    g ^= g >> 1;
    return g;
}

// square Gray QAM: label (gray^-1(row) << k/2) | gray^-1(col); row 0 = +max imag, col 0 = -max real
// (fundamental.py:697-777).  Decision-identical to demod_mindist away from exact ties.
template <typename T>
__device__ __forceinline__ int demod_qam_slicer(cx<T> r, T scale, int L, int half_bits) {
    const T lm1 = (T)(L - 1);
    T fj = floor((r.x * scale + lm1) * (T)0.5 + (T)0.5);
    T fi = floor((lm1 - r.y * scale) * (T)0.5 + (T)0.5);
    fj = fj < (T)0 ? (T)0 : (fj > lm1 ? lm1 : fj);
    fi = fi < (T)0 ? (T)0 : (fi > lm1 ? lm1 : fi);
    return (gray2binary8((int)fi) << half_bits) | gray2binary8((int)fj);
}

template <typename T>
__device__ __forceinline__ int demod_one(const ModemParams<T>& mp, const cx<T>* s_table, cx<T> r) {
    if (mp.method == MCLE_DEMOD_QAM_SLICER) return demod_qam_slicer<T>(r, mp.qam_scale, mp.qam_L, mp.half_bits);
    return demod_mindist<T>(s_table, mp.M, r);
}

// cooperative copy of the constellation into LDS (call before a __syncthreads())
template <typename T>
__device__ __forceinline__ void load_table(const ModemParams<T>& mp, cx<T>* s_table) {
    for (int m = threadIdx.x; m < mp.M; m += blockDim.x) s_table[m] = mp.g_table[m];
}

}  // namespace mcle
