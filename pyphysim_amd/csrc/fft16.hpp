// fft16.hpp -- the 1024-point transform as 16 x 16 x 4 with the two DFT-16 passes on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact f32, an fmaf chain per output), shared by the fused f32 pipelines
// (pipeline_mimo_mfma.hip, pipeline_siso_tdl.hip).  Reference transform: modulators/ofdm.py:394-466.
//
// FFT-1024 = 16 x 16 x 4.  A 16-point DFT of 16 independent groups is one real matrix product
//     [Re; Im](out) = [[Wr, -Wi], [Wi, Wr]] [Re; Im](in)
// after one radix-2 split on the VALU (s = x[e] + x[e+8], d = x[e] - x[e+8]: even outputs = DFT-8 of s, odd
// outputs = DFT-8 of d with W16^e folded into the matrix), i.e. 2 x (16x16 real) x (16 x 16 groups) = 8 MFMAs per
// (row, 16 groups) instead of 16.
//   DIF (transmit, inverse via the re<->im swap identity IDFT(x) = swap(DFT(swap(x)))):
//     P1  DFT-16 over n1 (positions 64 n1 + n2)        x W1024^{k1 n2}     wave w owns columns n2 in [16w, 16w+16)
//     P2  DFT-16 over m1 (positions 64 k1 + 4 m1 + m2) x W64^{j1 m2}       wave w owns rows k1 in [4w, 4w+4)
//     P3  DFT-4  over m2 (positions 64 k1 + 4 j1 + m2)                      thread = one butterfly, same rows
//     -> position 64 k1 + 4 j1 + j2 holds time sample k1 + 16 j1 + 256 j2
//   DIT (receive) is the transpose: P3' (DFT-4, x W64), P2' (DFT-16, x W1024), P1' (DFT-16) -> natural bin order.
//
// LDS: per row (antenna / realization slot) a re plane and an im plane of 1024 floats (im plane 16 dwords further: a
// wave's two planes hit complementary bank halves), position p stored at p ^ (f(p >> 6) << 2),
// f(k) = (k & 7) ^ ((k & 1) << 3): every access of the passes above is bank-conflict free
// (tests/test_fft16_layout.py replays all of them; the b128 stores of the middle stage are 2-way, below their own
// issue cost).
#pragma once
#include "common.hpp"

namespace mcle {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kF16N = 1024;
constexpr int kF16Plane = 1040;            // dwords from an antenna's re plane to its im plane (== 16 mod 32)
constexpr int kF16Ant = 2 * kF16Plane;     // dwords per antenna

__host__ __device__ __forceinline__ int f16_swz(int k) { return ((k & 7) ^ ((k & 1) << 3)) << 2; }
__host__ __device__ __forceinline__ int f16_pos(int p) { return p ^ f16_swz(p >> 6); }

__device__ __forceinline__ float dpp_swap1(float v) {   // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// The two real 16x16 matrices of the split DFT-16, as MFMA A operands: lane (row i = l & 15, k-group g = l >> 4),
// k-step t < 4 covers element e = 2t + (g >> 1), part g & 1; row i = 2u + part_out.
struct Dft16Mats {
    float ae[4], ao[4];
};
__device__ __forceinline__ Dft16Mats dft16_mats(const float2* __restrict__ g_tw, int lane) {
    Dft16Mats m;
    const int i = lane & 15, g = lane >> 4, u = i >> 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = 2 * t + (g >> 1);
        const float2 we = g_tw[(128 * e * u) & 1023];             // W8^{e u}
        const float2 wo = g_tw[(64 * e * (2 * u + 1)) & 1023];    // W16^{e (2u+1)}
        if ((i & 1) == 0) {
            m.ae[t] = (g & 1) ? -we.y : we.x;
            m.ao[t] = (g & 1) ? -wo.y : wo.x;
        } else {
            m.ae[t] = (g & 1) ? we.x : we.y;
            m.ao[t] = (g & 1) ? wo.x : wo.y;
        }
    }
    return m;
}

// One DFT-16 pass over one antenna's 16 groups: b[0..7] = this lane's operand values (element 2t + (g >> 1), part
// g & 1).  Returns out[x] = (re, im) of output 4g + x of group (lane & 15), x = 0..3.
__device__ __forceinline__ void dft16_mfma(const Dft16Mats& m, const float (&b)[8], float2 (&out)[4]) {
    f4 ce = {0.f, 0.f, 0.f, 0.f}, co = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ce = __builtin_amdgcn_mfma_f32_16x16x4f32(m.ae[t], b[t] + b[t + 4], ce, 0, 0, 0);
        co = __builtin_amdgcn_mfma_f32_16x16x4f32(m.ao[t], b[t] - b[t + 4], co, 0, 0, 0);
    }
    out[0] = make_float2(ce[0], ce[1]);   // k = 4g     (u = 2g,     even)
    out[1] = make_float2(co[0], co[1]);   // k = 4g + 1 (u = 2g,     odd)
    out[2] = make_float2(ce[2], ce[3]);   // k = 4g + 2 (u = 2g + 1, even)
    out[3] = make_float2(co[2], co[3]);   // k = 4g + 3
}

// complex product in the shape the backend lowers to v_pk_mul_f32 + v_pk_fma_f32 (two packed ops per product)
__device__ __forceinline__ float2 cmul_pk(float2 a, float2 w) {
    const float tx = a.x * w.x, ty = a.x * w.y;
    return make_float2(fmaf(-a.y, w.y, tx), fmaf(a.y, w.x, ty));
}

// The NA antennas (or realization slots) of one DFT-16 pass: all operand loads first (the pass is in place per wavefront, so
// nothing it stores is read again inside it), the 8 NA MFMAs as 2 NA independent accumulator chains, then twiddle and store.
template <int NA, typename LoadOff, typename StoreOff, typename Fill>
__device__ __forceinline__ void dft16_pass(float* s_d, int plane_g, const Dft16Mats& m, const float2 (&tw)[4],
                                           LoadOff ld, StoreOff st, Fill fill) {
    float b[NA][8];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < 8; ++t) b[a][t] = s_d[a * kF16Ant + plane_g + ld(t)];
    f4 ce[NA], co[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ce[a] = co[a] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            ce[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(m.ae[t], b[a][t] + b[a][t + 4], ce[a], 0, 0, 0);
            co[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(m.ao[t], b[a][t] - b[a][t + 4], co[a], 0, 0, 0);
        }
    fill();
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const float2 o[4] = {make_float2(ce[a][0], ce[a][1]), make_float2(co[a][0], co[a][1]),
                             make_float2(ce[a][2], ce[a][3]), make_float2(co[a][2], co[a][3])};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float2 v = cmul_pk(o[x], tw[x]);
            const int off = a * kF16Ant + st(x);
            s_d[off] = v.x;
            s_d[off + kF16Plane] = v.y;
        }
    }
}
template <typename LoadOff, typename StoreOff, typename Fill>
__device__ __forceinline__ void dft16_pass4(float* s_d, int plane_g, const Dft16Mats& m, const float2 (&tw)[4],
                                            LoadOff ld, StoreOff st, Fill fill) {
    dft16_pass<4>(s_d, plane_g, m, tw, ld, st, fill);
}

// inverse of ofdm_bin (fft.hpp): data index carried by FFT bin `bin`, or -1
__device__ __forceinline__ int ofdm_data_index(int bin, int n, int num_used) {
    if (num_used == n) return (bin + n / 2) & (n - 1);
    const int h = num_used / 2;
    if (bin >= n - h) return bin - (n - h);
    if (bin >= 1 && bin <= h) return h + bin - 1;
    return -1;
}

}  // namespace mcle
