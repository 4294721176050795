// Host build of csrc/bm_f64.hpp for tests/test_bm_f64_cpu.py: the same source the device compiles, with the storage /
// function qualifiers and the two hardware builtins replaced (the rsq estimate is degraded to float precision so that the
// Newton refinement is what the test exercises).
#include <cmath>
#include <cstddef>
#include <cstdint>
#define MCLE_BM_TABLE static const
#define MCLE_BM_FN static inline
#define MCLE_BM_RSQ(a) ((double)(1.0f / std::sqrt((float)(a))))
#define MCLE_BM_FMA(a, b, c) std::fma(a, b, c)
#define MCLE_BM_RINT(a) std::nearbyint(a)
#include "../../pyphysim_amd/csrc/bm_f64.hpp"

extern "C" {
void bm_neg_log_batch(const uint32_t* x0, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = mcle::bm_neg_log(x0[i]);
}
void bm_sqrt_batch(const double* a, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = mcle::bm_sqrt(a[i]);
}
void bm_sincos_rad_batch(const double* x, double* c, double* s, size_t n) {
    for (size_t i = 0; i < n; ++i) mcle::bm_sincos_rad(x[i], c[i], s[i]);
}
void bm_sincos_batch(const uint32_t* x1, double* c, double* s, size_t n) {
    for (size_t i = 0; i < n; ++i) mcle::bm_sincos(x1[i], c[i], s[i]);
}
}
