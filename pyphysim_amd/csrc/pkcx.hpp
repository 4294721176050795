// pkcx.hpp -- complex64 arithmetic on (re, im) register pairs as EXPLICIT v_pk_*_f32 instructions (gfx950 VOP3P).
// A complex multiply-add is two packed FMAs when the halves are routed by the instruction's own modifiers -- op_sel / op_sel_hi
// pick the low or high dword of a source pair for the low / high result, neg_lo / neg_hi negate a source for one half:
//     acc + a b        = acc + a.x (b.x, b.y) + a.y (-b.y, b.x)
//     acc + a conj(b)  = acc + b.x (a.x, a.y) + b.y (a.y, -a.x)
// The backend finds these forms for some source shapes and not for others: in the wavefront-per-antenna MIMO-TDL kernel
// (mimo_tdl_wave.hpp) it built the second factor with a v_xor / v_mov per product and multiplied complex numbers in four packed
// instructions + a move (measured: a third of the instructions of the channel stage and of the frequency-response stage).  These
// helpers fix the form; they are plain (non-volatile) asm statements, so the scheduler still moves them and places the waits.
// Rounding: each product term is ONE fused multiply-add, as in cfma(float2, ...) of common.hpp (four chained FMAs).
#pragma once
#include "common.hpp"

// libmcle is written for gfx950 only (Makefile: ARCH; mcle_ctx_create refuses any other device).  The packed-f32 VOP3P forms below
// exist on gfx90a / gfx942 / gfx950; a device pass for anything else stops here with a sentence instead of an assembler error in
// the middle of thirty translation units.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "libmcle: pkcx.hpp needs v_pk_fma_f32 / v_pk_mul_f32 (gfx90a, gfx942, gfx950); build with ARCH=gfx950"
#endif

namespace mcle {

typedef float pk2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ pk2 to_pk(float2 a) { return (pk2){a.x, a.y}; }
__device__ __forceinline__ float2 from_pk(pk2 a) { return make_float2(a.x, a.y); }

// acc + a b
__device__ __forceinline__ pk2 pk_cfma(pk2 a, pk2 b, pk2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));                                   // + a.x (b.x, b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));    // + a.y (-b.y, b.x)
    return acc;
}
// a b
__device__ __forceinline__ pk2 pk_cmul(pk2 a, pk2 b) {
    pk2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));                                             // a.x (b.x, b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(t) : "v"(a), "v"(b));
    return t;
}
// acc + a conj(b)
__device__ __forceinline__ pk2 pk_cfma_conj(pk2 a, pk2 b, pk2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));                                   // + (a.x, a.y) b.x
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));    // + (a.y, -a.x) b.y
    return acc;
}

// One (tap, [transmit antenna,] sample) step of the delay line of the wavefront kernels (siso_tdl_wave.hpp, mimo_tdl_wave.hpp): g = Horner(cc, xx) (a polynomial with complex coefficients in a real
// abscissa), y += g xv.  complex64: four v_pk_fma_f32 -- Horner one per order on the (re, im) pair, the complex multiply-add two
// (pkcx.hpp: op_sel / neg modifiers swap and negate the halves); written on clang vector types and explicit instructions because
// the backend does not form the packed Horner from scalar FMAs with wave-uniform coefficients (4 v_fma + 2 v_pk_fma + 3 v_mov per step).
template <int KO, bool ASM = true> __device__ __forceinline__ void chan_step(float2& y, const float2 (&cc)[KO + 1], float xx, float2 xv) {
    pk2 g = {cc[KO].x, cc[KO].y};
    const pk2 x2 = {xx, xx};
#pragma unroll
    for (int mm = KO - 1; mm >= 0; --mm) g = __builtin_elementwise_fma(g, x2, (pk2){cc[mm].x, cc[mm].y});
    if constexpr (ASM) {
        y = from_pk(pk_cfma(g, to_pk(xv), to_pk(y)));
    } else {
        pk2 acc = {y.x, y.y};
        acc = __builtin_elementwise_fma((pk2){g.x, g.x}, (pk2){xv.x, xv.y}, acc);
        acc = __builtin_elementwise_fma((pk2){-g.y, g.y}, (pk2){xv.y, xv.x}, acc);
        y = from_pk(acc);
    }
}
template <int KO, bool ASM = true> __device__ __forceinline__ void chan_step(double2& y, const double2 (&cc)[KO + 1], double xx, double2 xv) {
    double2 g = cc[KO];
#pragma unroll
    for (int mm = KO - 1; mm >= 0; --mm) {
        g.x = fma(g.x, xx, cc[mm].x);
        g.y = fma(g.y, xx, cc[mm].y);
    }
    y = cfma4(g, xv, y);
}

}  // namespace mcle
