// philox.hpp -- the "mcle-philox-v1" randomness contract, device side.
//
// Philox4x32-10 (Salmon et al. SC'11; rocRAND's philox4x32_10) as a pure function:
//   key = (seed_lo, seed_hi), counter = (block, stream, realization_lo, realization_hi)
// == rocrand_init(seed, subsequence = realization, offset = 4*(stream*2^32 + block)).
// Replaces the reference's global NumPy MT19937 draws (util/misc.py:327-355 randn_c,
// np.random.randint in apps/awgn_modulators/simulate_psk.py:65) on the fused GPU path; the
// injected-input operator kernels take the reference's own draws instead.
//
// Derived draws (the NumPy statement of the same rules lives in the test oracle):
//   symbols   n -> block n/16, word (n/4)%4, byte n%4, & (M-1)
//   uniform   i -> block i/4, word i%4, * 2^-32                         (double)
//   CN(0,1)   i -> block i/2, words (2(i%2), 2(i%2)+1) = (x0, x1):
//             sqrt(-ln((x0+.5) 2^-32)) * exp(2 pi j x1 2^-32)            (Box-Muller)
#pragma once
#include "common.hpp"
#include "bm_f64.hpp"

namespace mcle {

enum : uint32_t { STREAM_DATA = 0, STREAM_NOISE = 1, STREAM_CHAN = 2, STREAM_PHASE = 3 };

struct Words4 {
    uint32_t w[4];
};

__host__ __device__ __forceinline__ Words4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                         uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
#if defined(__HIP_DEVICE_COMPILE__)
        // one v_bitop3_b32 (a ^ b ^ c, truth table 0x96) per word: the backend otherwise emits two v_xor_b32 for most of them
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#else
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#endif
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Words4 o;
    o.w[0] = c0;
    o.w[1] = c1;
    o.w[2] = c2;
    o.w[3] = c3;
    return o;
}

struct Rng {
    uint32_t k0, k1, r0, r1;
    __host__ __device__ Rng(uint64_t seed, uint64_t realization)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), r0((uint32_t)realization),
          r1((uint32_t)(realization >> 32)) {}
    __host__ __device__ __forceinline__ Words4 block(uint32_t stream, uint32_t blk) const {
        return philox4x32_10(blk, stream, r0, r1, k0, k1);
    }
};

// ---- Box-Muller ---------------------------------------------------------------------------
// sigma = sqrt(variance of the complex sample); returns sigma * CN(0,1)
__device__ __forceinline__ float2 cn_from_words(uint32_t x0, uint32_t x1, float sigma) {
    const float u = fmaf((float)x0, 0x1p-32f, 0x1p-33f);
    const float v = (float)x1 * 0x1p-32f;  // revolutions
    // -ln(u) = -log2(u) * ln2 ; v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (input in turns)
    // sigma sqrt(-ln u) as sqrt((-ln 2 sigma^2) log2 u): the constant is loop-invariant, one multiply less per sample
    const float rad = __builtin_amdgcn_sqrtf((-0.69314718055994531f * sigma * sigma) * __builtin_amdgcn_logf(u));
    float2 z;
    z.x = rad * __builtin_amdgcn_cosf(v);
    z.y = rad * __builtin_amdgcn_sinf(v);
    return z;
}
// complex128: the table + polynomial forms of bm_f64.hpp (within one unit in the last place of the device libm's
// log / sincos / sqrt this replaced, at a quarter of the instructions)
__device__ __forceinline__ double2 cn_from_words(uint32_t x0, uint32_t x1, double sigma) {
    const double rad = sigma * bm_sqrt(bm_neg_log(x0));
    double s, c;
    bm_sincos(x1, c, s);
    double2 z;
    z.x = rad * c;
    z.y = rad * s;
    return z;
}

// the same with the tables of bm_f64.hpp read from the caller's LDS copy (bm_tables_to_lds)
__device__ __forceinline__ double2 cn_from_words_lds(uint32_t x0, uint32_t x1, double sigma, const double* s_bm) {
    const double rad = sigma * bm_sqrt(bm_neg_log(x0, s_bm));
    double s, c;
    bm_sincos(x1, c, s, s_bm + kBmLogLen, s_bm + kBmLogLen + kBmThetaLen);
    double2 z;
    z.x = rad * c;
    z.y = rad * s;
    return z;
}

// complex sample i of (stream): one Philox call, half of it used
template <typename T>
__device__ __forceinline__ cx<T> cn_sample(const Rng& rng, uint32_t stream, uint64_t i, T sigma) {
    const Words4 b = rng.block(stream, (uint32_t)(i >> 1));
    // (selects, not b.w[h]: a lane-dependent index into the four words makes the backend park them in SCRATCH -- a 16-byte store
    //  and a dependent load per sample inside the symbol walks, round 5)
    const bool hi = (i & 1) != 0;
    return cn_from_words(hi ? b.w[2] : b.w[0], hi ? b.w[3] : b.w[1], sigma);
}

// samples 2*blk and 2*blk+1 from one Philox call
template <typename T>
__device__ __forceinline__ void cn_pair(const Rng& rng, uint32_t stream, uint32_t blk, T sigma,
                                        cx<T>& z0, cx<T>& z1) {
    const Words4 b = rng.block(stream, blk);
    z0 = cn_from_words(b.w[0], b.w[1], sigma);
    z1 = cn_from_words(b.w[2], b.w[3], sigma);
}

// the same against the caller's LDS copy of the complex128 tables (complex64 has none and ignores it)
__device__ __forceinline__ void cn_pair_lds(const Rng& rng, uint32_t stream, uint32_t blk, float sigma, float2& z0,
                                            float2& z1, const double*) {
    cn_pair<float>(rng, stream, blk, sigma, z0, z1);
}
__device__ __forceinline__ void cn_pair_lds(const Rng& rng, uint32_t stream, uint32_t blk, double sigma, double2& z0,
                                            double2& z1, const double* s_bm) {
    const Words4 b = rng.block(stream, blk);
    z0 = cn_from_words_lds(b.w[0], b.w[1], sigma, s_bm);
    z1 = cn_from_words_lds(b.w[2], b.w[3], sigma, s_bm);
}

__device__ __forceinline__ double uniform_at(const Rng& rng, uint32_t stream, uint64_t i) {
    const Words4 b = rng.block(stream, (uint32_t)(i >> 2));
    const uint32_t lo = (i & 1) ? b.w[1] : b.w[0], hi = (i & 1) ? b.w[3] : b.w[2];      // (selects: see cn_sample)
    return (double)((i & 2) ? hi : lo) * 0x1p-32;
}

// symbol n of the data stream
__device__ __forceinline__ uint32_t symbol_at(const Rng& rng, uint64_t n, uint32_t mask) {
    const Words4 b = rng.block(STREAM_DATA, (uint32_t)(n >> 4));
    const uint32_t lo = (n & 4) ? b.w[1] : b.w[0], hi = (n & 4) ? b.w[3] : b.w[2];      // (selects: see cn_sample)
    return (((n & 8) ? hi : lo) >> ((n & 3) * 8)) & mask;
}

}  // namespace mcle
