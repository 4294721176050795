// fft.hpp -- workgroup-cooperative, in-place, in-LDS complex FFT for power-of-two sizes.
//
// Stands in for np.fft.fft / np.fft.ifft as used by OFDM.modulate / demodulate (reference
// modulators/ofdm.py:421-422,456-457).  Radix-4 butterflies with one trailing radix-2 stage when
// log2(N) is odd.  Two forms that chain WITHOUT any reordering pass:
//     fft_dif : natural-order input  -> digit-reversed output   (decimation in frequency)
//     fft_dit : digit-reversed input -> natural-order output    (decimation in time)
// so an OFDM link chains the two without a reordering pass, in either order: IFFT(dif) -> per-sample work
// on scrambled time samples (position p holds time index fft_index_of_pos(p)) -> FFT(dit), used where the
// channel is memoryless (config 4); or -- symbols scattered straight into digit-reversed bin positions --
// IFFT(dit) -> time samples in NATURAL order -> FFT(dif) -> bins gathered from digit-reversed positions, used
// by the tapped-delay-line kernels, whose x[m - d] gathers are then contiguous (conflict-free) LDS reads.
// `nf` independent transforms (antennas) sit side by side in LDS, `pitch` elements apart; the
// whole workgroup shares every stage.  Twiddles w[k] = exp(-2 pi i k / N) come from an LDS (or
// global) table of N entries built in double precision on the host.
#pragma once
#include "common.hpp"

namespace mcle {

template <int N> struct FftShape {
    static_assert(N >= 4 && (N & (N - 1)) == 0, "power of two >= 4");
    static constexpr int log2n() {
        int l = 0;
        for (int v = N; v > 1; v >>= 1) ++l;
        return l;
    }
    static constexpr int LOG2 = log2n();
    static constexpr int N4 = LOG2 / 2;     // radix-4 stages
    static constexpr bool HAS2 = LOG2 & 1;  // trailing radix-2 stage (span 1)
};

// natural index f  ->  position after fft_dif (where bin / sample f ends up)
template <int N> __host__ __device__ __forceinline__ int fft_pos_of_index(int f) {
    int pos = 0, size = N;
#pragma unroll
    for (int s = 0; s < FftShape<N>::N4; ++s) {
        size >>= 2;
        pos += (f & 3) * size;
        f >>= 2;
    }
    if (FftShape<N>::HAS2) pos += (f & 1);
    return pos;
}
// position p -> natural index held there after fft_dif (inverse of the above)
template <int N> __host__ __device__ __forceinline__ int fft_index_of_pos(int p) {
    int f = 0, size = N, mul = 1;
#pragma unroll
    for (int s = 0; s < FftShape<N>::N4; ++s) {
        size >>= 2;
        const int q = p / size;
        p -= q * size;
        f += q * mul;
        mul <<= 2;
    }
    if (FftShape<N>::HAS2) f += p * mul;
    return f;
}

// OFDM data position d in [0, num_used) -> FFT bin (reference modulators/ofdm.py:188-224): full
// band puts data k on bin (k + N/2) mod N; otherwise the first half rides the negative bins
// [N-h, N-1] and the second half the positive bins [1, h] (DC and band edges unused).
__host__ __device__ __forceinline__ int ofdm_bin(int d, int n, int num_used) {
    if (num_used == n) {   // n even here (num_used is); no power-of-two assumption
        const int t = d + n / 2;
        return t >= n ? t - n : t;
    }
    const int h = num_used / 2;
    return d < h ? n - h + d : 1 + (d - h);
}

// twiddle index (prod mod n): mask = n-1 for powers of two, -1 otherwise (uniform branch)
__host__ __device__ __forceinline__ int tw_index(int prod, int n, int mask) {
    return mask >= 0 ? (prod & mask) : (int)((unsigned)prod % (unsigned)n);
}
__host__ __forceinline__ int tw_mask_of(int n) { return (n & (n - 1)) == 0 ? n - 1 : -1; }
// sizes the radix-4 LDS kernels are instantiated for
__host__ __forceinline__ bool fft_is_radix4_size(int n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }
// N = N1 * N2 with N1 <= N2 the divisor pair closest to sqrt(N) (N1 = 1 for primes)
__host__ __forceinline__ void dft_any_split(int n, int* n1, int* n2) {
    int best = 1;
    for (int d = 1; d * d <= n; ++d)
        if (n % d == 0) best = d;
    *n1 = best;
    *n2 = n / best;
}

// LDS bank swizzle for the in-place radix-4 stages (8-byte elements: float2, or one plane of doubles).  8-byte accesses
// obey two bank rules on gfx950: a ds_read_b64 is served per half-wave of 32 lanes over 32 eight-byte slots, a ds_write_b64
// per 16 consecutive lanes over 16 slots.  Stages with span s < 64 touch s-element runs that are 4s apart, which piles 2
// (s = 16) or 4 (s = 4, 1) lanes on a slot.  Folding index bits 4..5 into bits 0..3 (twice) and bit 6 into bit 4 makes every
// stage's four accesses -- loads AND stores -- and every contiguous aligned run conflict free (checked exhaustively in
// tests/test_fft_layout.py and tests/test_f64_layout.py).  Until round 3 the swizzle was built for the read rule only
// (bits 5..6 into 0..3, bit 6 into 4): the stores of the spans 16, 4 and 1 were 2- to 4-way conflicted -- 12.8 extra LDS
// cycles per store instruction averaged over the five stages in the bank model, 0.29 of the LDS cycles of the
// frequency-selective MIMO kernel in the counters.
// (Round 3 also tried folding bits 7..9 in -- bit 0 ^= b7, bit 1 ^= b6, bit 2 ^= b9, bit 3 ^= b8 -- which additionally makes
// the digit-reversed bin gathers / symbol scatters of the tapped-delay-line kernels conflict free (56 extra LDS cycles per
// read in the bank model otherwise).  Measured: f1 6.20 -> 6.05 ms, but the generic kernels that recompute their addresses
// per stage lost 2-9 % to the longer address arithmetic (config 4 VALU kernel 1.93 -> 1.97 ms, complex128 generic 5.37 ->
// 5.86 ms): the LDS is not what limits them.  Not kept.)
// SWZ = false keeps the linear layout (operator kernels with global-memory twiddles).
template <bool SWZ> __host__ __device__ __forceinline__ int lds_swz(int e) {
    if (!SWZ) return e;
    return e ^ (((e >> 4) & 3) * 5) ^ (((e >> 6) & 1) << 4);
}

// The four positions e0, e0 + s, e0 + 2s, e0 + 3s of a radix-4 butterfly (e0 = g 4s + k, k < s: the two bits of q s are
// clear in e0, so e0 + q s = e0 ^ q s) through the swizzle, which is linear over XOR (bit extractions and shifts only):
// swz(e0 ^ q s) = swz(e0) ^ swz(q s), the second term a compile-time constant per stage -- one swizzle and three XORs where
// four swizzles cost 28 VALU instructions per butterfly.
template <bool SWZ> __host__ __device__ __forceinline__ void lds_swz_r4(int e0, int s, int& i0, int& i1, int& i2, int& i3) {
    i0 = lds_swz<SWZ>(e0);
    i1 = i0 ^ lds_swz<SWZ>(s);
    i2 = i0 ^ lds_swz<SWZ>(2 * s);
    i3 = i0 ^ lds_swz<SWZ>(3 * s);
}

template <typename T, bool INV> __device__ __forceinline__ cx<T> tw_get(const cx<T>* tw, int i) {
    cx<T> w = tw[i];
    if (INV) w.y = -w.y;
    return w;
}
// multiply by -i (forward) or +i (inverse)
template <typename T, bool INV> __device__ __forceinline__ cx<T> rot(cx<T> a) {
    return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x);
}

// Synchronisation between two radix-4 stages.  With a compile-time workgroup size that is a multiple of 64 the
// s butterflies that share a 4s-point region sit on s consecutive threads, and no region straddles a wavefront
// once s <= 64: such a stage only has to be ordered against the SAME wavefront's earlier LDS traffic, which
// executes in order -- a fence for the compiler and the counters, no workgroup barrier.  (`s` is the span of
// the stage whose s threads exchange data: the producing stage in DIF, the consuming one in DIT.)
template <int NTHREADS> __device__ __forceinline__ void fft_stage_sync(int s) {
    if (NTHREADS > 0 && NTHREADS % 64 == 0 && s <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

// ---- decimation in frequency: natural -> digit-reversed ----------------------------------------
// Caller has written s_data and synchronised.  Returns synchronised.
// NTHREADS > 0: the workgroup size is a compile-time constant, so the per-stage butterfly loops
// unroll (independent LDS round trips in flight; with nf*N/4 a multiple of NTHREADS and N/4 ==
// NTHREADS every lane runs the SAME butterfly of each transform and shares its twiddles).
// FRESH: re-derive the butterfly addresses from an opaque thread index in every stage.  Without it the
// optimiser hoists all stages' addresses out of the caller's realization loop -- faster as long as they fit
// in registers (C3 / C4), a source of spills in kernels with fatter phases around the transforms.
template <typename T, int N, bool INV, int NTHREADS = 0, bool SWZ = false, bool FRESH = false>
__device__ __forceinline__ void fft_dif(cx<T>* s_data, int nf, int pitch, const cx<T>* tw) {
    constexpr int NB = N / 4;
    const int nthreads = NTHREADS > 0 ? NTHREADS : (int)blockDim.x;
    int s = N / 4;
#pragma unroll
    for (int st = 0; st < FftShape<N>::N4; ++st, s >>= 2) {
        const int twstep = N / (4 * s);
#pragma unroll 4
        for (int b = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x; b < nf * NB; b += nthreads) {
            const int f = b / NB, bb = b - f * NB;
            const int k = bb & (s - 1), g = bb / s;
            cx<T>* p = s_data + f * pitch;
            const int e0 = g * 4 * s + k;
            int i0, i1, i2, i3;
            lds_swz_r4<SWZ>(e0, s, i0, i1, i2, i3);
            const cx<T> x0 = p[i0], x1 = p[i1], x2 = p[i2], x3 = p[i3];
            const cx<T> a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = rot<T, INV>(csub(x1, x3));
            cx<T> y0 = cadd(a0, a2), y1 = cadd(a1, a3), y2 = csub(a0, a2), y3 = csub(a1, a3);
            if (s > 1) {
                y1 = cmul(y1, tw_get<T, INV>(tw, k * twstep));
                y2 = cmul(y2, tw_get<T, INV>(tw, 2 * k * twstep));
                y3 = cmul(y3, tw_get<T, INV>(tw, 3 * k * twstep));
            }
            p[i0] = y0;
            p[i1] = y1;
            p[i2] = y2;
            p[i3] = y3;
        }
        if (st + 1 < FftShape<N>::N4)
            fft_stage_sync<NTHREADS>(s);   // the next stage reads what THIS stage's s-thread groups wrote
        else
            __syncthreads();               // leave (or enter the radix-2 stage) workgroup-synchronised
    }
    if (FftShape<N>::HAS2) {
        for (int b = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x; b < nf * (N / 2); b += nthreads) {
            const int f = b / (N / 2), bb = b - f * (N / 2);
            cx<T>* p = s_data + f * pitch;
            const int i0 = lds_swz<SWZ>(2 * bb), i1 = lds_swz<SWZ>(2 * bb + 1);
            const cx<T> x0 = p[i0], x1 = p[i1];
            p[i0] = cadd(x0, x1);
            p[i1] = csub(x0, x1);
        }
        __syncthreads();
    }
}

// ---- decimation in time: digit-reversed -> natural ------------------------------------------------
template <typename T, int N, bool INV, int NTHREADS = 0, bool SWZ = false, bool FRESH = false>
__device__ __forceinline__ void fft_dit(cx<T>* s_data, int nf, int pitch, const cx<T>* tw) {
    constexpr int NB = N / 4;
    const int nthreads = NTHREADS > 0 ? NTHREADS : (int)blockDim.x;
    if (FftShape<N>::HAS2) {
        for (int b = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x; b < nf * (N / 2); b += nthreads) {
            const int f = b / (N / 2), bb = b - f * (N / 2);
            cx<T>* p = s_data + f * pitch;
            const int i0 = lds_swz<SWZ>(2 * bb), i1 = lds_swz<SWZ>(2 * bb + 1);
            const cx<T> x0 = p[i0], x1 = p[i1];
            p[i0] = cadd(x0, x1);
            p[i1] = csub(x0, x1);
        }
        __syncthreads();
    }
    int s = FftShape<N>::HAS2 ? 2 : 1;
#pragma unroll
    for (int st = 0; st < FftShape<N>::N4; ++st, s <<= 2) {
        const int twstep = N / (4 * s);
#pragma unroll 4
        for (int b = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x; b < nf * NB; b += nthreads) {
            const int f = b / NB, bb = b - f * NB;
            const int k = bb & (s - 1), g = bb / s;
            cx<T>* p = s_data + f * pitch;
            const int e0 = g * 4 * s + k;
            int i0, i1, i2, i3;
            lds_swz_r4<SWZ>(e0, s, i0, i1, i2, i3);
            cx<T> u0 = p[i0], u1 = p[i1], u2 = p[i2], u3 = p[i3];
            if (s > 1) {
                u1 = cmul(u1, tw_get<T, INV>(tw, k * twstep));
                u2 = cmul(u2, tw_get<T, INV>(tw, 2 * k * twstep));
                u3 = cmul(u3, tw_get<T, INV>(tw, 3 * k * twstep));
            }
            const cx<T> a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = rot<T, INV>(csub(u1, u3));
            p[i0] = cadd(a0, a2);
            p[i1] = cadd(a1, a3);
            p[i2] = csub(a0, a2);
            p[i3] = csub(a1, a3);
        }
        if (st + 1 < FftShape<N>::N4)
            fft_stage_sync<NTHREADS>(4 * s);   // the next stage's 4s-thread groups read what was written here
        else
            __syncthreads();
    }
}


// ---- register-resident twiddles ------------------------------------------------------------------
// With NTHREADS == N/4 every thread runs ONE butterfly position per stage (the same in each of the nf
// transforms), so its three twiddles per stage are loop invariants of the caller's realization loop:
// fft_twiddle_regs() loads them once, the *_r transforms take them from registers (no LDS twiddle reads).
template <int N> struct FftTwRegs {
    static constexpr int STAGES = FftShape<N>::N4;
};
template <typename T, int N, int NTHREADS>
__device__ __forceinline__ void fft_twiddle_regs(const cx<T>* tw, cx<T> (&r)[FftShape<N>::N4][3]) {
    static_assert(NTHREADS == N / 4, "one butterfly per thread and stage");
    int s = N / 4;
#pragma unroll
    for (int st = 0; st < FftShape<N>::N4; ++st, s >>= 2) {      // DIF stage order: spans N/4, N/16, ...
        const int k = (int)threadIdx.x & (s - 1), twstep = N / (4 * s);
        r[st][0] = tw[k * twstep];
        r[st][1] = tw[2 * k * twstep];
        r[st][2] = tw[3 * k * twstep];
    }
}
template <typename T, bool INV> __device__ __forceinline__ cx<T> tw_reg(cx<T> w) {
    if (INV) w.y = -w.y;
    return w;
}
template <typename T, int N, bool INV, int NTHREADS, bool SWZ, bool FRESH = false>
__device__ __forceinline__ void fft_dif_r(cx<T>* s_data, int nf, int pitch, const cx<T> (&twr)[FftShape<N>::N4][3]) {
    static_assert(NTHREADS == N / 4 && !FftShape<N>::HAS2, "radix-4 only, one butterfly per thread and stage");
    int s = N / 4;
#pragma unroll
    for (int st = 0; st < FftShape<N>::N4; ++st, s >>= 2) {
        const int bb = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x;
        const int k = bb & (s - 1), g = bb / s;
        const int e0 = g * 4 * s + k;
        int i0, i1, i2, i3;
        lds_swz_r4<SWZ>(e0, s, i0, i1, i2, i3);
        const cx<T> w1 = tw_reg<T, INV>(twr[st][0]), w2 = tw_reg<T, INV>(twr[st][1]), w3 = tw_reg<T, INV>(twr[st][2]);
#pragma unroll 4
        for (int f = 0; f < nf; ++f) {
            cx<T>* p = s_data + f * pitch;
            const cx<T> x0 = p[i0], x1 = p[i1], x2 = p[i2], x3 = p[i3];
            const cx<T> a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = rot<T, INV>(csub(x1, x3));
            cx<T> y0 = cadd(a0, a2), y1 = cadd(a1, a3), y2 = csub(a0, a2), y3 = csub(a1, a3);
            if (s > 1) {
                y1 = cmul(y1, w1);
                y2 = cmul(y2, w2);
                y3 = cmul(y3, w3);
            }
            p[i0] = y0;
            p[i1] = y1;
            p[i2] = y2;
            p[i3] = y3;
        }
        if (st + 1 < FftShape<N>::N4)
            fft_stage_sync<NTHREADS>(s);
        else
            __syncthreads();
    }
}
template <typename T, int N, bool INV, int NTHREADS, bool SWZ, bool FRESH = false>
__device__ __forceinline__ void fft_dit_r(cx<T>* s_data, int nf, int pitch, const cx<T> (&twr)[FftShape<N>::N4][3]) {
    static_assert(NTHREADS == N / 4 && !FftShape<N>::HAS2, "radix-4 only, one butterfly per thread and stage");
    int s = 1;
#pragma unroll
    for (int st = 0; st < FftShape<N>::N4; ++st, s <<= 2) {
        const int bb = FRESH ? opaque((int)threadIdx.x) : (int)threadIdx.x;
        const int k = bb & (s - 1), g = bb / s;
        const int e0 = g * 4 * s + k;
        int i0, i1, i2, i3;
        lds_swz_r4<SWZ>(e0, s, i0, i1, i2, i3);
        // DIT stage with span s uses the registers of the DIF stage with the same span
        const int rs = FftShape<N>::N4 - 1 - st;
        const cx<T> w1 = tw_reg<T, INV>(twr[rs][0]), w2 = tw_reg<T, INV>(twr[rs][1]), w3 = tw_reg<T, INV>(twr[rs][2]);
#pragma unroll 4
        for (int f = 0; f < nf; ++f) {
            cx<T>* p = s_data + f * pitch;
            cx<T> u0 = p[i0], u1 = p[i1], u2 = p[i2], u3 = p[i3];
            if (s > 1) {
                u1 = cmul(u1, w1);
                u2 = cmul(u2, w2);
                u3 = cmul(u3, w3);
            }
            const cx<T> a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = rot<T, INV>(csub(u1, u3));
            p[i0] = cadd(a0, a2);
            p[i1] = cadd(a1, a3);
            p[i2] = csub(a0, a2);
            p[i3] = csub(a1, a3);
        }
        if (st + 1 < FftShape<N>::N4)
            fft_stage_sync<NTHREADS>(4 * s);
        else
            __syncthreads();
    }
}

}  // namespace mcle
