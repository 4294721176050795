// qam_pack.hpp -- the packed square-QAM slicer of the f32 fused pipelines (four decisions at a time).
#pragma once
#include "modem.hpp"

namespace mcle {

typedef float f4q __attribute__((ext_vector_type(4)));

// Packed square-QAM slicer.  A label byte is (binary row << hb) | binary column with binary = gray^-1(level)
// (reference modulators/fundamental.py:697-777; demod_qam_slicer in modem.hpp).  Working in the LEVEL domain --
// the sent bytes are stored as levels, level = b ^ (b >> 1) per field -- the four decisions of a subcarrier are
// rounded, clamped and packed by v_cvt_pk_u8_f32 (round to nearest even, saturating: probed on the device,
// scripts/experiments/cvt_probe.hip), compared with one XOR, and the bit errors follow from one field-wise prefix
// XOR of the difference word: popcount(gray^-1(a) ^ gray^-1(b)) = popcount(gray^-1(a ^ b)).
struct QamPack {
    float sc, off, lm1;     // level = round(+-coordinate * sc + off), clamped to [0, lm1]
    uint32_t m1, m2;        // per byte: bits of both fields that have a neighbour 1 / 2 places up inside the field
    int hb;
};
__device__ __forceinline__ QamPack qam_pack(const ModemParams<float>& mp) {
    QamPack q;
    q.lm1 = (float)(mp.qam_L - 1);
    q.sc = 0.5f * mp.qam_scale;
    q.off = 0.5f * q.lm1;
    q.hb = mp.half_bits;
    const uint32_t fm = (1u << q.hb) - 1u;
    q.m1 = (((fm >> 1) | ((fm >> 1) << q.hb)) & 0xFFu) * 0x01010101u;
    q.m2 = (((fm >> 2) | ((fm >> 2) << q.hb)) & 0xFFu) * 0x01010101u;
    return q;
}
__device__ __forceinline__ uint32_t labels_to_levels(uint32_t w, const QamPack& q) { return w ^ ((w >> 1) & q.m1); }
// level word of the four estimates (re[a], im[a]), a = byte index
__device__ __forceinline__ uint32_t qam_levels4(const f4q& re, const f4q& im, const QamPack& q) {
    uint32_t wj = 0u, wi = 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        wj = __builtin_amdgcn_cvt_pk_u8_f32(fminf(fmaf(re[a], q.sc, q.off), q.lm1), a, wj);
        wi = __builtin_amdgcn_cvt_pk_u8_f32(fminf(fmaf(im[a], -q.sc, q.off), q.lm1), a, wi);
    }
    return (wi << q.hb) | wj;
}
// x = decided levels ^ sent levels of four symbols -> (+symbol errors, +bit errors)
__device__ __forceinline__ void qam_count4(uint32_t x, const QamPack& q, unsigned& se, unsigned& be) {
    const uint32_t t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    se += __popc(t);
    uint32_t y = x ^ ((x >> 1) & q.m1);
    y ^= (y >> 2) & q.m2;
    be += __popc(y);
}



}  // namespace mcle
