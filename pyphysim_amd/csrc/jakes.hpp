// jakes.hpp -- one ray of the Jakes sum-of-sinusoids model, shared by the operator kernel and the
// fused pipelines.  Reference: channels/fading_generators.py:459-467 (time axis) and :519-522
//   h = sqrt(1/L) * sum_l exp(1j * (2*pi*Fd*cos(phi_l)*t + psi_l)).
//
// Phases reach 2*pi*Fd*t ~ 6e4 rad (Fd = 100 Hz, t <= 100 s), far beyond what f32 argument
// reduction survives, so the phase is always formed in f64:
//   f64 path (parity):  w = (2*pi*Fd)*cos(phi) [rad/s];  x = fl(fl(w*t) + psi)  -- the reference's
//                       own evaluation order, no FMA contraction -- then cos / sin of that double (bm_sincos_rad:
//                       within 1.2e-16 of the exact values, like NumPy's own).
//   f32 path (speed):   w = Fd*cos(phi) [turns/s], psi/(2*pi) [turns]; x = fma(w, t, psi) in f64,
//                       fract(x) -> f32 -> v_sin_f32 / v_cos_f32 (their input unit is turns).
#pragma once
#include <cmath>

#include "common.hpp"
#include "bm_f64.hpp"

namespace mcle {

inline double jakes_w(int dtype, double Fd, double phi) {
    const double pi = 3.141592653589793238462643383279502884;
    if (dtype == MCLE_F64) return 2 * pi * Fd * std::cos(phi);  // ((2*pi)*Fd)*cos(phi), as NumPy evaluates it
    return Fd * std::cos(phi);
}
inline double jakes_psi(int dtype, double psi) {
    const double pi = 3.141592653589793238462643383279502884;
    return dtype == MCLE_F64 ? psi : psi / (2 * pi);
}

// t_i = t0 + i*dt with the two roundings of numpy.arange (start + i*delta)
__device__ __forceinline__ double jakes_time(double t0, double dt, double i) {
    return __dadd_rn(t0, __dmul_rn(i, dt));
}

template <typename T> __device__ __forceinline__ cx<T> jakes_ray(double w, double psi, double t);

template <> __device__ __forceinline__ double2 jakes_ray<double>(double w, double psi, double t) {
    const double x = __dadd_rn(__dmul_rn(w, t), psi);
    double s, c;
    if (fabs(x) <= 0x1p24) bm_sincos_rad(x, c, s);       // table + polynomial form (bm_f64.hpp): a quarter of libm's instructions
    else sincos(x, &s, &c);                              // phases beyond 1.6e7 rad (t > 7 h at Fd = 100 Hz): the library routine
    double2 r;
    r.x = c;
    r.y = s;
    return r;
}
template <> __device__ __forceinline__ float2 jakes_ray<float>(double w, double psi, double t) {
    const double x = fma(w, t, psi);
    const float v = (float)__builtin_amdgcn_fract(x);     // x - floor(x) in one v_fract_f64
    float2 r;
    r.x = __builtin_amdgcn_cosf(v);
    r.y = __builtin_amdgcn_sinf(v);
    return r;
}

}  // namespace mcle
