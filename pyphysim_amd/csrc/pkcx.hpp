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

}  // namespace mcle
