// bm_f64.hpp -- the complex128 Box-Muller of the mcle-philox-v1 contract (philox.hpp), written for gfx950's f64 datapath.
//
//   z = sigma * sqrt(-ln((x0 + 0.5) 2^-32)) * exp(j * fl(2 pi_d * x1 2^-32))       (oracle: oracle/philox.py cnormal)
//
// The library forms (log, sincos, sqrt of the device libm) cost ~180 double-precision instructions per sample and were
// more than half of every complex128 pipeline (round-3 profile of k_run_mimo_ofdm<double>: 6 500 VALU instructions per
// wavefront and realization, 3 400 of them here).  The arguments are structured -- u is an odd multiple of 2^-33, the
// angle is a 32-bit fraction of a turn -- so both functions reduce to one small table look-up and a short polynomial:
//
//   -ln u   u = m 2^e, m in [1, 2) (folded to [0.75, 1.5) so that u -> 1 meets c = 1 with ln c = 0 exactly);
//           j = the nearest of 65 nodes c_j, r = (m - c_j) / c_j exactly representable difference times a rounded
//           reciprocal, |r| <= 2^-7;  ln u = e ln 2 + ln c_j + log1p(r), log1p by its degree-7 series (next term 2^-59).
//           Measured against x87 extended precision over 4e6 words incl. the 1e5 largest: relative error <= 2.3e-16
//           (NumPy's own log: 1.2e-16), i.e. <= 3e-16 absolute on sqrt(-ln u).
//   sincos  NumPy evaluates cos / sin of the DOUBLE ang = fl(2 pi_d v); the nodes theta_k = fl(k fl(2 pi / 128)) are
//           doubles too, so r = ang - theta_k is exact (Sterbenz), |r| <= 0.0246, and the table holds cos / sin of the
//           double theta_k: rotation by the degree-7 / degree-6 polynomials of r.  <= 1.2e-16 absolute against the
//           extended-precision value of cos(ang), sin(ang).
//   sqrt    v_rsq_f64 + one coupled Newton step + two residual corrections (argument range [2e-10, 23]: no scaling).
//
// ~47 f64 instructions + 12 integer ones + 3 table reads per sample.  The tables are 4.1 KiB: static device arrays (served
// from L1) by default, or a kernel's own LDS copy where the latency matters (pipeline_mimo_planar.hip).  tests/test_bm_f64_cpu.py compiles this header
// for the host and checks it word by word against NumPy; the -m gpu parity tests then hold every complex128 pipeline's
// per-realization error counts equal to the oracle's.
#pragma once
#include <cstdint>

#ifndef MCLE_BM_TABLE
#define MCLE_BM_TABLE static __device__ const
#endif
#ifndef MCLE_BM_FN
#define MCLE_BM_FN __device__ __forceinline__
#endif
#ifndef MCLE_BM_RSQ
#define MCLE_BM_RSQ(a) __builtin_amdgcn_rsq(a)
#endif
#ifndef MCLE_BM_FMA
#define MCLE_BM_FMA(a, b, c) __builtin_fma(a, b, c)
#endif
#ifndef MCLE_BM_RINT
#define MCLE_BM_RINT(a) __builtin_rint(a)
#endif

#include "bm_tables.hpp"

namespace mcle {

constexpr int kBmLogLen = 65 * 2, kBmThetaLen = 129, kBmTrigLen = 129 * 2;     // doubles: 4.1 KiB in all

// A kernel's own LDS copy of the three tables, laid out [kBmLog | kBmTheta | kBmTrig] (call before a barrier)
constexpr int kBmLdsDoubles = kBmLogLen + kBmThetaLen + kBmTrigLen;
#ifdef __HIPCC__
__device__ __forceinline__ void bm_tables_to_lds(double* s_bm, int tid, int nthreads) {
    for (int i = tid; i < kBmLogLen; i += nthreads) s_bm[i] = kBmLog[i];
    for (int i = tid; i < kBmThetaLen; i += nthreads) s_bm[kBmLogLen + i] = kBmTheta[i];
    for (int i = tid; i < kBmTrigLen; i += nthreads) s_bm[kBmLogLen + kBmThetaLen + i] = kBmTrig[i];
}
#endif

// -ln((x0 + 0.5) 2^-32)
MCLE_BM_FN double bm_neg_log(uint32_t x0, const double* tlog = kBmLog) {
    const double ud = (double)x0 + 0.5;                                   // exact: u 2^32
    const uint64_t bits = __builtin_bit_cast(uint64_t, ud);
    const uint32_t hi = (uint32_t)(bits >> 32);
    const uint32_t mant = hi & 0xFFFFFu;
    const uint32_t j = (mant + 0x2000u) >> 14;                            // nearest node, 0 .. 64
    const bool fold = j >= 32u;                                           // m >= 1.5: use m / 2 and e + 1
    const uint32_t ebase = fold ? 0x3FE00000u : 0x3FF00000u;
    const double m = __builtin_bit_cast(double, ((uint64_t)(ebase | mant) << 32) | (uint64_t)(uint32_t)bits);
    const double c = __builtin_bit_cast(double, (uint64_t)(ebase + (j << 14)) << 32);    // (1 + j/64) [/ 2]; j = 64 -> 1
    const int e = (int)(hi >> 20) - (1023 + 32) + (fold ? 1 : 0);
    const double inv_c = tlog[2 * j], lnc = tlog[2 * j + 1];
    const double r = (m - c) * inv_c;
    const double r2 = r * r;
    // log1p(r) = r + r^2 (-1/2 + r/3 + r^2 ((-1/4 + r/5) + r^2 (-1/6 + r/7)))
    const double a0 = MCLE_BM_FMA(r, 1.0 / 3.0, -0.5);
    const double a1 = MCLE_BM_FMA(r, 0.2, -0.25);
    const double a2 = MCLE_BM_FMA(r, 1.0 / 7.0, -1.0 / 6.0);
    const double q = MCLE_BM_FMA(r2, MCLE_BM_FMA(r2, a2, a1), a0);
    const double small = MCLE_BM_FMA(r2, q, r);
    const double big = MCLE_BM_FMA((double)e, 0x1.62e42fefa39efp-1, lnc);    // e ln 2 + ln c_j, one rounding
    return -(big + small);
}

// sqrt(a), a in [2e-10, 23]
MCLE_BM_FN double bm_sqrt(double a) {
    const double y = MCLE_BM_RSQ(a);
    double g = a * y, h = 0.5 * y;
    const double r = MCLE_BM_FMA(-h, g, 0.5);
    g = MCLE_BM_FMA(g, r, g);
    h = MCLE_BM_FMA(h, r, h);
    double d = MCLE_BM_FMA(-g, g, a);
    g = MCLE_BM_FMA(d, h, g);
    d = MCLE_BM_FMA(-g, g, a);
    return MCLE_BM_FMA(d, h, g);
}

// cos / sin of the double fl(2 pi_d * x1 2^-32)
MCLE_BM_FN void bm_sincos(uint32_t x1, double& c, double& s, const double* ttheta = kBmTheta, const double* ttrig = kBmTrig) {
    const double ang = (double)x1 * 0x1.921fb54442d18p-30;                // (2 pi_d) 2^-32: fl(.) == NumPy's 2.0*np.pi*(x1*2**-32)
    const uint32_t k = ((x1 >> 24) + 1u) >> 1;                            // nearest node, 0 .. 128
    const double ct = ttrig[2 * k], st = ttrig[2 * k + 1];
    const double r = ang - ttheta[k];                                     // exact
    const double s2 = r * r;
    double p = MCLE_BM_FMA(s2, -1.0 / 5040.0, 1.0 / 120.0);
    p = MCLE_BM_FMA(s2, p, -1.0 / 6.0);
    const double sr = MCLE_BM_FMA(r * s2, p, r);                          // sin r
    double q = MCLE_BM_FMA(s2, -1.0 / 720.0, 1.0 / 24.0);
    q = MCLE_BM_FMA(s2, q, -0.5);
    const double cm = s2 * q;                                             // cos r - 1
    c = ct + MCLE_BM_FMA(-st, sr, ct * cm);
    s = st + MCLE_BM_FMA(ct, sr, st * cm);
}

// The same with the node angle and its cos / sin in ONE 32-byte entry {cos, sin, theta, -} (a kernel's own LDS copy, built by
// bm_trig_packed_to_lds): one address for both reads of a sample (variant measured in round 4, pipeline_mimo_planar.hip)
MCLE_BM_FN void bm_sincos_packed(uint32_t x1, double& c, double& s, const double* tpk) {
    const double ang = (double)x1 * 0x1.921fb54442d18p-30;
    const uint32_t k = ((x1 >> 24) + 1u) >> 1;
    const double ct = tpk[4 * k], st = tpk[4 * k + 1];
    const double r = ang - tpk[4 * k + 2];
    const double s2 = r * r;
    double p = MCLE_BM_FMA(s2, -1.0 / 5040.0, 1.0 / 120.0);
    p = MCLE_BM_FMA(s2, p, -1.0 / 6.0);
    const double sr = MCLE_BM_FMA(r * s2, p, r);
    double q = MCLE_BM_FMA(s2, -1.0 / 720.0, 1.0 / 24.0);
    q = MCLE_BM_FMA(s2, q, -0.5);
    const double cm = s2 * q;
    c = ct + MCLE_BM_FMA(-st, sr, ct * cm);
    s = st + MCLE_BM_FMA(ct, sr, st * cm);
}
#ifdef __HIPCC__
constexpr int kBmPackedDoubles = 129 * 4;
__device__ __forceinline__ void bm_trig_packed_to_lds(double* s_pk, int tid, int nthreads) {
    for (int i = tid; i < kBmThetaLen; i += nthreads) {
        s_pk[4 * i] = kBmTrig[2 * i];
        s_pk[4 * i + 1] = kBmTrig[2 * i + 1];
        s_pk[4 * i + 2] = kBmTheta[i];
        s_pk[4 * i + 3] = 0.0;
    }
}
#endif

// cos / sin of a general double x (radians), |x| <= 2^24: the Jakes ray phases of the complex128 kernels (2 pi Fd t cos(phi) + psi
// reaches 6e4 rad).  x / (2 pi) in two words -- p = fl(x C1) with its exact residual by one FMA, plus x C2 -- gives the turn
// count to 1e-33; k = rint(4 p) picks the quadrant, p - k/4 is an exact subtraction, and that residual (+ the two small words)
// times 2 pi, |r| <= pi/4, goes through the Taylor polynomials of sin (degree 17) and cos (degree 18); the quadrant is a swap
// and two sign flips (exact).  No table: a first version looked the nearest of 128 node angles up in a 2 KiB table and was
// SLOWER than the library routine inside the fused kernels (config 2, complex128: 58.8 vs 38.0 ms per 16 384 realizations) --
// eight dependent global gathers per symbol at two wavefronts per SIMD.  <= 1.7e-16 absolute against x87 extended precision
// over +-1e5 rad (tests/test_bm_f64_cpu.py); NumPy's own values are within 1.1e-16 of the same.
// (No v_fract_f64 on p: p - floor(p) rounds for negative p -- 3.5e-16 rad near odd multiples of pi/2.)
MCLE_BM_FN void bm_sincos_rad(double x, double& c, double& s) {
    const double C1 = 0x1.45f306dc9c883p-3;                               // fl(1 / (2 pi))
    const double C2 = -0x1.6b01ec5417056p-57;                             // 1 / (2 pi) - C1
    const double p = x * C1;
    const double e = MCLE_BM_FMA(x, C1, -p) + x * C2;                      // turns = p + e
    const double kd = MCLE_BM_RINT(p * 4.0);                              // |kd| < 2^24
    const int k = (int)kd;
    const double rt = MCLE_BM_FMA(kd, -0.25, p) + e;                      // residual in turns: exact difference (|.| <= 1/8) + e
    const double r = rt * 0x1.921fb54442d18p+2;                           // x 2 pi
    const double z = r * r;
    // sin r = r + r z (S1 + z (S2 + ... )), cos r = 1 + z (C1 + z (C2 + ...)): Taylor coefficients (-1)^n / (2n+1)!, / (2n)!
    double ps = MCLE_BM_FMA(z, 1.0 / 355687428096000.0, -1.0 / 1307674368000.0);       // 1/17!, -1/15!
    ps = MCLE_BM_FMA(z, ps, 1.0 / 6227020800.0);                                      // 1/13!
    ps = MCLE_BM_FMA(z, ps, -1.0 / 39916800.0);                                       // -1/11!
    ps = MCLE_BM_FMA(z, ps, 1.0 / 362880.0);                                          // 1/9!
    ps = MCLE_BM_FMA(z, ps, -1.0 / 5040.0);                                           // -1/7!
    ps = MCLE_BM_FMA(z, ps, 1.0 / 120.0);                                             // 1/5!
    ps = MCLE_BM_FMA(z, ps, -1.0 / 6.0);                                              // -1/3!
    const double sr = MCLE_BM_FMA(r * z, ps, r);
    double pc = MCLE_BM_FMA(z, -1.0 / 6402373705728000.0, 1.0 / 20922789888000.0);     // -1/18!, 1/16!
    pc = MCLE_BM_FMA(z, pc, -1.0 / 87178291200.0);                                    // -1/14!
    pc = MCLE_BM_FMA(z, pc, 1.0 / 479001600.0);                                       // 1/12!
    pc = MCLE_BM_FMA(z, pc, -1.0 / 3628800.0);                                        // -1/10!
    pc = MCLE_BM_FMA(z, pc, 1.0 / 40320.0);                                           // 1/8!
    pc = MCLE_BM_FMA(z, pc, -1.0 / 720.0);                                            // -1/6!
    pc = MCLE_BM_FMA(z, pc, 1.0 / 24.0);                                              // 1/4!
    pc = MCLE_BM_FMA(z, pc, -0.5);
    const double cr = MCLE_BM_FMA(z, pc, 1.0);
    // quadrant k mod 4: (c, s) = (cr, sr), (-sr, cr), (-cr, -sr), (sr, -cr)
    const double a = (k & 1) ? sr : cr, b = (k & 1) ? cr : sr;
    c = ((k + 1) & 2) ? -a : a;
    s = (k & 2) ? -b : b;
}

}  // namespace mcle
