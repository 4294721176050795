// fft_r16.hpp -- radix-16 register passes of a 1024-point transform held by ONE WAVEFRONT, on planar LDS samples (a re plane
// and an im plane of N scalars), for float and double.  Written in round 4 inside pipeline_mimo_planar.hip (config 4: wavefront =
// antenna); shared since with pipeline_siso_tdl_wave.hip (config 3: wavefront = realization).
#pragma once
#include <type_traits>
#include <utility>

#include "common.hpp"
#include "fft.hpp"

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
    static __device__ __forceinline__ C fma(C h, C x, C acc) { return cfma(h, x, acc); }
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
template <typename T, bool INV, bool DIT, int WHICH, bool EXACT = false, bool REGIN = false, bool REGOUT = false>
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
                v[1][q] = r16_root<T, INV, q * 1>(CxOps<T>::template mulw<INV>(v[1][q], t1[0]));
                v[2][q] = r16_root<T, INV, q * 2>(CxOps<T>::template mulw<INV>(v[2][q], t1[1]));
                v[3][q] = r16_root<T, INV, q * 3>(CxOps<T>::template mulw<INV>(v[3][q], t1[2]));
            }
        });
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            r4_inplace<T, INV>(v[m][0], v[m][1], v[m][2], v[m][3]);
#pragma unroll
            for (int q = 1; q < 4; ++q) v[m][q] = CxOps<T>::template mulw<INV>(v[m][q], t2[q - 1]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int q = 1; q < 4; ++q) v[m][q] = CxOps<T>::template mulw<INV>(v[m][q], t2[q - 1]);
            r4_inplace<T, INV>(v[m][0], v[m][1], v[m][2], v[m][3]);
        }
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (EXACT) {
                v[1][q] = CxOps<T>::template mulw<INV>(v[1][q], tw1(qc, std::integral_constant<int, 1>()));
                v[2][q] = CxOps<T>::template mulw<INV>(v[2][q], tw1(qc, std::integral_constant<int, 2>()));
                v[3][q] = CxOps<T>::template mulw<INV>(v[3][q], tw1(qc, std::integral_constant<int, 3>()));
            } else {
                v[1][q] = r16_root<T, INV, q * 1>(CxOps<T>::template mulw<INV>(v[1][q], t1[0]));
                v[2][q] = r16_root<T, INV, q * 2>(CxOps<T>::template mulw<INV>(v[2][q], t1[1]));
                v[3][q] = r16_root<T, INV, q * 3>(CxOps<T>::template mulw<INV>(v[3][q], t1[2]));
            }
            r4_inplace<T, INV>(v[0][q], v[1][q], v[2][q], v[3][q]);
        });
    }
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
template <typename T, bool INV, bool WITH_C = true, bool EXACT = false, bool REGIN = false>
__device__ __forceinline__ void r16_dif(T* pr, T* pi, int lane, const R16Tw64<T>& tw, const cx<T>* __restrict__ g_tw = nullptr,
                                        const cx<T>* vin = nullptr) {
    int gi = opaque(lane);
    r16_pass<T, INV, false, 0, EXACT, REGIN>(pr, pi, lds_swz16f(gi), tw, g_tw, gi, vin);
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

}  // namespace mcle
