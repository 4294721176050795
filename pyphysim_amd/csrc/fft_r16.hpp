// fft_r16.hpp -- radix-16 register passes of a 1024-point transform held by ONE WAVEFRONT, on planar LDS samples (a re plane
// and an im plane of N scalars), for float and double.  Written in round 4 inside pipeline_mimo_planar.hip (config 4: wavefront =
// antenna); shared since with pipeline_siso_tdl_wave.hip (config 3: wavefront = realization).
#pragma once
#include <type_traits>
#include <utility>

#include "common.hpp"
#include "fft.hpp"
#include "pkcx.hpp"

namespace mcle {

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, COUNT - 1>)
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int COUNT, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, COUNT>());
}

// Complex arithmetic of the transforms, the channel and the decode in one place.  A PACKED complex64 specialization (a value =
// one 64-bit register pair, v_pk_add / v_pk_mul / v_pk_fma_f32 with op_sel swizzles and sign modifiers as inline asm: a radix-4
// butterfly in 8 instructions instead of 16, a twiddle product in 2 instead of 4) was measured in round 4 and is kept as
// scripts/experiments/f32_packed_cx_r04.patch: 23 % fewer VALU instructions, no gain in time at the benchmark geometry (4.88 ->
// 5.06 ms per 262 144 realizations; -18 % .. +15 % over the family) -- on gfx950 a packed f32 op issues in 4.3 - 4.6 cycles
// against 2.7 - 3.0 for v_add / v_mul / v_fma_f32 (scripts/experiments/f32_rates.hip -> profiles/r04/f32_rates.txt), so packing
// buys 1.2 - 1.4 x per flop at best, and the asm blocks cost the scheduler its view of the latencies.
template <typename T> struct CxOps {
    using C = cx<T>;
    static __device__ __forceinline__ C add(C a, C b) { return cadd(a, b); }
    static __device__ __forceinline__ C sub(C a, C b) { return csub(a, b); }
    template <bool CONJ> static __device__ __forceinline__ C mulw(C a, C w) {       // a w  or  a conj(w)
        if (CONJ) w.y = -w.y;
        return cmul(a, w);
    }
    // acc + h x: four chained FMAs (common.hpp), or -- PK, complex64 only -- two packed FMAs (pkcx.hpp)
    template <bool PK = false> static __device__ __forceinline__ C fma(C h, C x, C acc) {
        if constexpr (PK && sizeof(T) == 4) return from_pk(pk_cfma(to_pk(h), to_pk(x), to_pk(acc)));
        else return cfma4(h, x, acc);
    }
    // y0 = (u0 + u2) + (u1 + u3), y2 = (u0 + u2) - (u1 + u3), y1 / y3 = (u0 - u2) +/- r (u1 - u3), r = -i (forward), +i (INV)
    template <bool INV> static __device__ __forceinline__ void bfly4(C u0, C u1, C u2, C u3, C& y0, C& y1, C& y2, C& y3) {
        const C a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = rot<T, INV>(csub(u1, u3));
        y0 = cadd(a0, a2);
        y1 = cadd(a1, a3);
        y2 = csub(a0, a2);
        y3 = csub(a1, a3);
    }
};
// ---- radix-16 passes for N = 1024 = 16 x 16 x 4, ONE TRANSFORM PER WAVEFRONT (variant 4 of the 4 x 4 geometry) ----------------
// Wavefront f owns antenna f's transform; lane gi keeps a 16-point group in registers across two radix-4 layers:
//   pass A = spans 256, 64: elements gi + 64 q + 256 m
//   pass B = spans 16, 4:   elements 64 (gi / 4) + gi % 4 + 4 q + 16 m
//   pass C = span 1:        elements 16 gi + 4 c + m          (four plain radix-4 butterflies)
// = three LDS round trips per transform instead of five, and -- a transform never leaves its wavefront -- no workgroup
// barrier inside a transform (only the channel and the decode, which need every antenna of a position, are fenced).
// The layer-1 twiddle w^((k + 64 q) m) is applied as w^(k m) (a register) times the constant 16th root w^(64 q m).
// (The complex64 form of this was measured in round 1 at the 168-register bound and lost to spills,
// scripts/experiments/radix16_fft.patch; complex128 at two workgroups per CU has 256 registers per lane.)
// LDS swizzle of this variant: index bits 4..8 folded into bits 0..4, bit 9 into bit 4 as well.  Linear over XOR; meets the
// 32-lane read rule AND the 16-lane store rule of 8-byte accesses for every shape of the three passes, the fused middle
// stage, the channel's position pairs, scatter and decode (tests/test_f64_layout.py derives it: the lane bits of each
// shape must map to independent slot bits).
__host__ __device__ __forceinline__ int lds_swz16f(int e) { return e ^ ((e >> 4) & 31) ^ (((e >> 9) & 1) << 4); }

template <typename T> struct R16Tw64 {
    cx<T> a1[3], a2[3], b1[3], b2[3];     // w^(k m), w^(4 k q) | w^(16 k4 m), w^(64 k4 q);  m, q = 1..3; k = lane, k4 = lane mod 4
};
template <typename T> __device__ __forceinline__ R16Tw64<T> load_r16_tw(const cx<T>* __restrict__ g_tw, int lane) {
    R16Tw64<T> r;
    const int k = lane & 63, k4 = k & 3;
#pragma unroll
    for (int j = 1; j <= 3; ++j) {
        r.a1[j - 1] = g_tw[k * j];
        r.a2[j - 1] = g_tw[4 * k * j];
        r.b1[j - 1] = g_tw[16 * k4 * j];
        r.b2[j - 1] = g_tw[64 * k4 * j];
    }
    return r;
}
// v times exp(-2 pi i n / 16) (forward) or its conjugate (inverse), n = q m in {0, 1, 2, 3, 4, 6, 9}
template <typename T, bool INV, int NN> __device__ __forceinline__ cx<T> r16_root(cx<T> v) {
    constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
    if constexpr (NN == 0) return v;
    else if constexpr (NN == 4) return rot<T, INV>(v);
    else {
        constexpr double re = NN == 1 ? c1 : NN == 2 ? h : NN == 3 ? s1 : NN == 6 ? -h : -c1;
        constexpr double im = NN == 1 ? -s1 : NN == 2 ? -h : NN == 3 ? -c1 : NN == 6 ? -h : s1;
        return CxOps<T>::template mulw<false>(v, mk<T>((T)re, (T)(INV ? -im : im)));
    }
}
template <typename T, bool INV> __device__ __forceinline__ void r4_inplace(cx<T>& x0, cx<T>& x1, cx<T>& x2, cx<T>& x3) {
    CxOps<T>::template bfly4<INV>(x0, x1, x2, x3, x0, x1, x2, x3);
}
__device__ __forceinline__ void r16_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// a 16-point register pass: WHICH = 0 (pass A: offsets 64 q + 256 m, twiddles a1 / a2), 1 (pass B: 4 q + 16 m, b1 / b2).
// DIF: butterflies over m, twiddle, butterflies over q, twiddle.  DIT: the mirror image, twiddles first.
// EXACT: the nine layer-1 twiddles of q >= 1, w^((k + 64 q) m) / w^(16 (k4 + 4 q) m), fetched from the (L1-resident) table at the
// top of the pass instead of formed as register twiddle x constant 16th root: eight complex multiplications less per pass, and
// the pass becomes the radix-4 stages' arithmetic operation for operation (bit-identical outputs).
// REGIN: the sixteen inputs come from the caller's registers (vin[q + 4 m] = element QS q + MS m) instead of the planes.
// REGOUT: the sixteen outputs go to the caller's registers (vout[q + 4 m]) instead of the planes.
// BAR: a workgroup barrier between the pass's arithmetic and its stores -- for a caller whose planes are still being READ by the
// other wavefronts of the workgroup when the pass starts (mimo_tdl_wave.hpp: the receive transform's first pass takes its inputs
// from registers while the other receive antennas finish their delay-line reads of this antenna's time signal).
// UNIT: every lane twiddle is 1 (t1 = t2 = 1: a 16-point transform of sixteen CONSECUTIVE elements, the constant 16th roots only) --
// the second register pass of a 256-point transform (pipeline_mimo_qw.hip)
template <typename T, bool INV, bool DIT, int WHICH, bool EXACT = false, bool REGIN = false, bool REGOUT = false, bool BAR = false,
          bool UNIT = false>
__device__ __forceinline__ void r16_pass(T* pr, T* pi, int base_slot, const R16Tw64<T>& tw,
                                         const cx<T>* __restrict__ g_tw = nullptr, int kidx = 0, const cx<T>* vin = nullptr,
                                         cx<T>* vout = nullptr) {
    constexpr int QS = WHICH == 0 ? 64 : 4, MS = WHICH == 0 ? 256 : 16;
    const cx<T>* t1 = WHICH == 0 ? tw.a1 : tw.b1;
    const cx<T>* t2 = WHICH == 0 ? tw.a2 : tw.b2;
    [[maybe_unused]] cx<T> tq[3][3];                      // [q - 1][m - 1]
    if constexpr (EXACT) {
#pragma unroll
        for (int q = 1; q < 4; ++q)
#pragma unroll
            for (int m = 1; m < 4; ++m)
                tq[q - 1][m - 1] = g_tw[(WHICH == 0 ? (kidx + 64 * q) * m : 16 * (kidx + 4 * q) * m) & 1023];
    }
    auto tw1 = [&](auto qc, auto mc) -> cx<T> {           // layer-1 twiddle of (q, m), m >= 1 (mulw conjugates it for the inverse)
        constexpr int q = decltype(qc)::value, m = decltype(mc)::value;
        if constexpr (EXACT && q > 0) return tq[q - 1][m - 1];
        else return t1[m - 1];
    };
    cx<T> v[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (REGIN) {
                v[m][q] = vin[q + 4 * m];
            } else {
                const int sl = base_slot ^ lds_swz16f(QS * q + MS * m);      // base and offsets occupy disjoint bits: XOR == add
                v[m][q] = mk<T>(pr[sl], pi[sl]);
            }
        }
    if constexpr (!DIT) {
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            r4_inplace<T, INV>(v[0][q], v[1][q], v[2][q], v[3][q]);
            if constexpr (EXACT) {
                v[1][q] = CxOps<T>::template mulw<INV>(v[1][q], tw1(qc, std::integral_constant<int, 1>()));
                v[2][q] = CxOps<T>::template mulw<INV>(v[2][q], tw1(qc, std::integral_constant<int, 2>()));
                v[3][q] = CxOps<T>::template mulw<INV>(v[3][q], tw1(qc, std::integral_constant<int, 3>()));
            } else {
                v[1][q] = r16_root<T, INV, q * 1>(UNIT ? v[1][q] : CxOps<T>::template mulw<INV>(v[1][q], t1[0]));
                v[2][q] = r16_root<T, INV, q * 2>(UNIT ? v[2][q] : CxOps<T>::template mulw<INV>(v[2][q], t1[1]));
                v[3][q] = r16_root<T, INV, q * 3>(UNIT ? v[3][q] : CxOps<T>::template mulw<INV>(v[3][q], t1[2]));
            }
        });
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            r4_inplace<T, INV>(v[m][0], v[m][1], v[m][2], v[m][3]);
#pragma unroll
            for (int q = 1; q < 4; ++q) if (!UNIT) v[m][q] = CxOps<T>::template mulw<INV>(v[m][q], t2[q - 1]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int q = 1; q < 4; ++q) if (!UNIT) v[m][q] = CxOps<T>::template mulw<INV>(v[m][q], t2[q - 1]);
            r4_inplace<T, INV>(v[m][0], v[m][1], v[m][2], v[m][3]);
        }
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (EXACT) {
                v[1][q] = CxOps<T>::template mulw<INV>(v[1][q], tw1(qc, std::integral_constant<int, 1>()));
                v[2][q] = CxOps<T>::template mulw<INV>(v[2][q], tw1(qc, std::integral_constant<int, 2>()));
                v[3][q] = CxOps<T>::template mulw<INV>(v[3][q], tw1(qc, std::integral_constant<int, 3>()));
            } else {
                v[1][q] = r16_root<T, INV, q * 1>(UNIT ? v[1][q] : CxOps<T>::template mulw<INV>(v[1][q], t1[0]));
                v[2][q] = r16_root<T, INV, q * 2>(UNIT ? v[2][q] : CxOps<T>::template mulw<INV>(v[2][q], t1[1]));
                v[3][q] = r16_root<T, INV, q * 3>(UNIT ? v[3][q] : CxOps<T>::template mulw<INV>(v[3][q], t1[2]));
            }
            r4_inplace<T, INV>(v[0][q], v[1][q], v[2][q], v[3][q]);
        });
    }
    if constexpr (BAR) __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (REGOUT) {
                vout[q + 4 * m] = v[m][q];
            } else {
                const int sl = base_slot ^ lds_swz16f(QS * q + MS * m);
                pr[sl] = v[m][q].x;
                pi[sl] = v[m][q].y;
            }
        }
}
// pass C: the four span-1 butterflies of elements 16 gi + 4 c + m (no twiddles; DIF and DIT share the add / sub network)
template <typename T, bool INV> __device__ __forceinline__ void r16_pass_c(T* pr, T* pi, int gi) {
    const int base_slot = lds_swz16f(16 * gi);
    cx<T> v[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int sl = base_slot ^ lds_swz16f(4 * c + m);
            v[c][m] = mk<T>(pr[sl], pi[sl]);
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r4_inplace<T, INV>(v[c][0], v[c][1], v[c][2], v[c][3]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int sl = base_slot ^ lds_swz16f(4 * c + m);
            pr[sl] = v[c][m].x;
            pi[sl] = v[c][m].y;
        }
    }
}
// natural -> digit-reversed (the arrangement of the radix-4 DIF stages) / digit-reversed -> natural; one wavefront, one antenna
// vin != nullptr (REGIN): pass A takes element gi + 64 q + 256 m of the input from vin[q + 4 m] (the caller's registers)
template <typename T, bool INV, bool WITH_C = true, bool EXACT = false, bool REGIN = false, bool BAR = false>
__device__ __forceinline__ void r16_dif(T* pr, T* pi, int lane, const R16Tw64<T>& tw, const cx<T>* __restrict__ g_tw = nullptr,
                                        const cx<T>* vin = nullptr) {
    int gi = opaque(lane);
    r16_pass<T, INV, false, 0, EXACT, REGIN, false, BAR>(pr, pi, lds_swz16f(gi), tw, g_tw, gi, vin);
    r16_wave_sync();
    gi = opaque(lane);
    r16_pass<T, INV, false, 1, EXACT>(pr, pi, lds_swz16f(64 * (gi >> 2) + (gi & 3)), tw, g_tw, gi & 3);
    if constexpr (WITH_C) {
        r16_wave_sync();
        r16_pass_c<T, INV>(pr, pi, opaque(lane));
    }
}
// vout != nullptr (REGOUT): the last pass (A) leaves element gi + 64 q + 256 m of the output in vout[q + 4 m] and stores nothing
template <typename T, bool INV, bool WITH_C = true, bool EXACT = false, bool REGOUT = false>
__device__ __forceinline__ void r16_dit(T* pr, T* pi, int lane, const R16Tw64<T>& tw, const cx<T>* __restrict__ g_tw = nullptr,
                                        cx<T>* vout = nullptr) {
    if constexpr (WITH_C) {
        r16_pass_c<T, INV>(pr, pi, opaque(lane));
        r16_wave_sync();
    }
    int gi = opaque(lane);
    r16_pass<T, INV, true, 1, EXACT>(pr, pi, lds_swz16f(64 * (gi >> 2) + (gi & 3)), tw, g_tw, gi & 3);
    r16_wave_sync();
    gi = opaque(lane);
    r16_pass<T, INV, true, 0, EXACT, false, REGOUT>(pr, pi, lds_swz16f(gi), tw, g_tw, gi, nullptr, vout);
}


// ---- radix-4 stages on planar samples (any power-of-two size; shared by the planar MIMO family and the wavefront kernels) ----
// LDS position of element e of a plane of doubles: the 8-byte-slot swizzle of fft.hpp (conflict free for the loads and the
// stores of every stage of this kernel: the legs of the radix-4 butterflies at every span, the trailing radix-2 stage, the
// channel's position pairs, scatter and decode, at every size of the family; tests/test_f64_layout.py replays all of them).
__host__ __device__ __forceinline__ int lds_swz64(int e) { return lds_swz<true>(e); }

// Radix-4 stage spans of an N-point transform in DIF order: N/4, N/16, ... down to 1 (N = 4^k) or 2 (N = 2 4^k, then one
// radix-2 stage on adjacent pairs closes the DIF / opens the DIT transform, as in fft.hpp).
template <int N> struct F64Shape {
    static constexpr int N4 = FftShape<N>::N4;
    static constexpr bool HAS2 = FftShape<N>::HAS2;
    static constexpr int NB = N / 4;                              // radix-4 butterfly positions of one antenna
    static constexpr int span(int st) { return (N / 4) >> (2 * st); }       // DIF stage st
    static constexpr int first_dit_span = HAS2 ? 2 : 1;
};

// The three twiddles of a thread's butterfly position at the radix-4 stages that have any (every span > 1):
// w[j][q] = W^{(q+1) k N/(4s)}, k = position mod s.  A DIF stage and the DIT stage of the same span use the same values
// (conjugated for the inverse transform); in the 256-thread form of N = 1024 they live in 48 registers for the whole
// kernel -- fetched per stage from the global table they sat on the critical path of every stage (three dependent
// ~600-cycle loads at two wavefronts per SIMD).
template <typename T, int N> struct TwRegs64 {
    cx<T> w[F64Shape<N>::N4][3];
};
template <typename T, int N> __device__ __forceinline__ TwRegs64<T, N> load_tw64(const cx<T>* __restrict__ g_tw, int bb) {
    TwRegs64<T, N> t;
#pragma unroll
    for (int j = 0; j < F64Shape<N>::N4; ++j) {
        const int s = F64Shape<N>::span(j), k = bb & (s - 1), ts = N / (4 * s);
#pragma unroll
        for (int q = 0; q < 3; ++q) t.w[j][q] = g_tw[(q + 1) * k * ts];
    }
    return t;
}

// the three twiddles of butterfly position bb at span S, fetched from the (L1-resident) table
template <typename T, int N, int S> __device__ __forceinline__ void stage_tw_fetch(const cx<T>* __restrict__ g_tw, int bb, cx<T> (&w)[3]) {
    constexpr int ts = N / (4 * S);
    const int k = bb & (S - 1);
    w[0] = g_tw[k * ts];
    w[1] = g_tw[2 * k * ts];
    w[2] = g_tw[3 * k * ts];
}

// one radix-4 butterfly position of AH antennas, planar LDS.
// DIF (INV: the transmit IFFT): butterfly, then twiddle; DIT (forward FFT): twiddle, then butterfly.
// pre: twiddles fetched ahead by the caller (forward transform: a DIT stage multiplies FIRST, so a fetch issued inside
// the stage sits on its critical path; issued one stage early it hides behind that stage's butterflies)
template <typename T, int N, bool DIF, bool INV, int S, int AH, bool TWR, bool NOSTORE = false>
__device__ __forceinline__ void r4_stage_planar(T* s_d, const TwRegs64<T, N>& tw, const cx<T>* __restrict__ g_tw, int bb,
                                                const cx<T>* pre = nullptr) {
    constexpr int s = S;
    const int k = bb & (s - 1), g = bb / s;
    const int e0 = g * 4 * s + k;
    int i0, i1, i2, i3;
    lds_swz_r4<true>(e0, s, i0, i1, i2, i3);           // one swizzle + three XORs with per-stage constants (fft.hpp)
    cx<T> w1 = mk<T>(1, 0), w2 = w1, w3 = w1;
    if (s > 1) {
        if constexpr (TWR) {                          // the thread's twiddles are registers
            constexpr int j = (FftShape<N>::LOG2 - FftShape<4 * S>::LOG2) / 2;     // DIF stage index of span S
            w1 = tw.w[j][0];
            w2 = tw.w[j][1];
            w3 = tw.w[j][2];
        } else if (pre != nullptr) {
            w1 = pre[0];
            w2 = pre[1];
            w3 = pre[2];
        } else {                                      // from the L1-resident table, per stage
            constexpr int ts = N / (4 * s);
            w1 = g_tw[k * ts];
            w2 = g_tw[2 * k * ts];
            w3 = g_tw[3 * k * ts];
        }
    }
    T xr[AH][4], xi[AH][4];
#pragma unroll
    for (int a = 0; a < AH; ++a) {
        const T* pr = s_d + (2 * a) * N;
        const T* pi = pr + N;
        xr[a][0] = pr[i0]; xr[a][1] = pr[i1]; xr[a][2] = pr[i2]; xr[a][3] = pr[i3];
        xi[a][0] = pi[i0]; xi[a][1] = pi[i1]; xi[a][2] = pi[i2]; xi[a][3] = pi[i3];
    }
#pragma unroll
    for (int a = 0; a < AH; ++a) {
        cx<T> u0 = mk<T>(xr[a][0], xi[a][0]), u1 = mk<T>(xr[a][1], xi[a][1]),
                u2 = mk<T>(xr[a][2], xi[a][2]), u3 = mk<T>(xr[a][3], xi[a][3]);
        if (!DIF && s > 1) {
            u1 = CxOps<T>::template mulw<INV>(u1, w1);
            u2 = CxOps<T>::template mulw<INV>(u2, w2);
            u3 = CxOps<T>::template mulw<INV>(u3, w3);
        }
        cx<T> y0, y1, y2, y3;
        CxOps<T>::template bfly4<INV>(u0, u1, u2, u3, y0, y1, y2, y3);
        if (DIF && s > 1) {
            y1 = CxOps<T>::template mulw<INV>(y1, w1);
            y2 = CxOps<T>::template mulw<INV>(y2, w2);
            y3 = CxOps<T>::template mulw<INV>(y3, w3);
        }
        if constexpr (NOSTORE) {                      // timing bound only (MCLE_OPT_F64_VARIANT): computed, not stored
            asm volatile("" ::"v"(y0.x), "v"(y0.y), "v"(y1.x), "v"(y1.y), "v"(y2.x), "v"(y2.y), "v"(y3.x), "v"(y3.y));
            continue;
        }
        T* pr = s_d + (2 * a) * N;
        T* pi = pr + N;
        pr[i0] = y0.x; pr[i1] = y1.x; pr[i2] = y2.x; pr[i3] = y3.x;
        pi[i0] = y0.y; pi[i1] = y1.y; pi[i2] = y2.y; pi[i3] = y3.y;
    }
}

// The radix-2 stage of N = 2 4^k (last of the DIF, first of the DIT transform; no twiddles): pairs (2p, 2p + 1).  Thread
// bb takes the two pairs inside ITS four consecutive positions 4 bb .. 4 bb + 3 -- the access pattern of the span-1
// radix-4 stage, and the points the lane pair (bb, bb ^ 1) exchanged in the span-2 stage next to it, so that the stage
// is ordered against its neighbour by the wavefront's own in-order LDS traffic (no workgroup barrier).
template <typename T, int N, int AH> __device__ __forceinline__ void r2_stage_planar(T* s_d, int bb) {
    int i0, i1, i2, i3;
    lds_swz_r4<true>(4 * bb, 1, i0, i1, i2, i3);
#pragma unroll
    for (int a = 0; a < AH; ++a) {
        T* pr = s_d + (2 * a) * N;
        T* pi = pr + N;
        const T r0 = pr[i0], r1 = pr[i1], r2 = pr[i2], r3 = pr[i3];
        const T m0 = pi[i0], m1 = pi[i1], m2 = pi[i2], m3 = pi[i3];
        pr[i0] = r0 + r1; pr[i1] = r0 - r1; pr[i2] = r2 + r3; pr[i3] = r2 - r3;
        pi[i0] = m0 + m1; pi[i1] = m0 - m1; pi[i2] = m2 + m3; pi[i3] = m2 - m3;
    }
}


// ---- an N-point transform held by ONE WAVEFRONT in radix-4 stages: N / 256 butterfly positions per lane and stage, every
// exchange wave-local (fences, no barrier); planes re = pr, im = pr + N, swizzle lds_swz64.  The sizes the radix-16 passes
// above do not cover (they are 1024 only): 256, 512, 2048 in the one-realization-per-wavefront kernels. ----
// TWR: the twiddles of the lane's REP butterfly positions live in registers (twr[rep] = load_tw64(g_tw, lane + 64 rep), loaded
// once per kernel) instead of being fetched from the table stage by stage -- an in-order wavefront waits for every such fetch,
// and a 256-point transform is too short to hide them.
template <typename T, int N, bool INV, bool TWR = false>
__device__ __forceinline__ void wave_fft_dif(T* pr, const cx<T>* __restrict__ g_tw, int lane, const TwRegs64<T, N>* twr = nullptr) {
    using SH = F64Shape<N>;
    constexpr int REP = SH::NB / 64;
    static_assert(REP >= 1, "one wavefront: N >= 256");
    TwRegs64<T, N> none;                                           // (unused without TWR: the stages fetch from the table)
    static_for<SH::N4>([&](auto stc) {
        constexpr int S = SH::span(decltype(stc)::value);
        const int gi = opaque(lane);
#pragma unroll
        for (int rep = 0; rep < REP; ++rep)
            r4_stage_planar<T, N, true, INV, S, 1, TWR>(pr, TWR ? twr[rep] : none, g_tw, gi + 64 * rep);
        r16_wave_sync();
    });
    if constexpr (SH::HAS2) {
        const int gi = opaque(lane);
#pragma unroll
        for (int rep = 0; rep < REP; ++rep) r2_stage_planar<T, N, 1>(pr, gi + 64 * rep);
        r16_wave_sync();
    }
}
template <typename T, int N, bool INV, bool TWR = false>
__device__ __forceinline__ void wave_fft_dit(T* pr, const cx<T>* __restrict__ g_tw, int lane, const TwRegs64<T, N>* twr = nullptr) {
    using SH = F64Shape<N>;
    constexpr int REP = SH::NB / 64;
    TwRegs64<T, N> none;
    if constexpr (SH::HAS2) {
        const int gi = opaque(lane);
#pragma unroll
        for (int rep = 0; rep < REP; ++rep) r2_stage_planar<T, N, 1>(pr, gi + 64 * rep);
        r16_wave_sync();
    }
    static_for<SH::N4>([&](auto stc) {
        constexpr int S = SH::span(SH::N4 - 1 - decltype(stc)::value);
        const int gi = opaque(lane);
#pragma unroll
        for (int rep = 0; rep < REP; ++rep)
            r4_stage_planar<T, N, false, INV, S, 1, TWR>(pr, TWR ? twr[rep] : none, g_tw, gi + 64 * rep);
        r16_wave_sync();
    });
}

}  // namespace mcle
