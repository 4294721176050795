// common.hpp -- shared host/device helpers of libmcle (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mcle.h"

namespace mcle {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define MCLE_HIP(expr)                                                                    \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::mcle::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                              __FILE__, __LINE__);                                        \
            return MCLE_E_HIP;                                                            \
        }                                                                                 \
    } while (0)

#define MCLE_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::mcle::set_error(__VA_ARGS__);     \
            return MCLE_E_INVAL;                \
        }                                       \
    } while (0)

#define MCLE_LAUNCH_CHECK() MCLE_HIP(hipGetLastError())

// wavefronts per SIMD the complex64 symbol walks k_ia_link and k_mimo_flat_link (one wavefront per workgroup) are bounded for:
// 4 = 128 registers with 6 / 4 of them spilled, 3 = 168 and none -- four is the faster one on the GPU (config 5: 3.44 against
// 3.33e8 realizations/s, flat MIMO 1.57 against 1.71 ms; profiles/r05/walk_occupancy_ab.log).  k_bd_link<float> runs at three.
#ifndef MCLE_F32_WALK_WAVES
#define MCLE_F32_WALK_WAVES 4
#endif

// ---- complex arithmetic on (re, im) pairs -------------------------------------------------
template <typename T> struct cx_of;
template <> struct cx_of<float> { using type = float2; };
template <> struct cx_of<double> { using type = double2; };
template <typename T> using cx = typename cx_of<T>::type;

template <typename T> __host__ __device__ __forceinline__ cx<T> mk(T re, T im) {
    cx<T> r;
    r.x = re;
    r.y = im;
    return r;
}
template <typename C> __host__ __device__ __forceinline__ C cadd(C a, C b) {
    a.x += b.x;
    a.y += b.y;
    return a;
}
template <typename C> __host__ __device__ __forceinline__ C csub(C a, C b) {
    a.x -= b.x;
    a.y -= b.y;
    return a;
}
template <typename C> __host__ __device__ __forceinline__ C cmul(C a, C b) {
    C r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
// a * conj(b)
template <typename C> __host__ __device__ __forceinline__ C cmulc(C a, C b) {
    C r;
    r.x = a.x * b.x + a.y * b.y;
    r.y = a.y * b.x - a.x * b.y;
    return r;
}
// acc + a*b
template <typename C> __host__ __device__ __forceinline__ C cfma(C a, C b, C acc) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
    return acc;
}
// f32: four chained FMAs -- the shape the backend turns into two v_pk_fma_f32 (op_sel / neg modifiers); the
// product-then-add form above costs v_pk_mul + v_pk_fma + v_mov + v_pk_add.  f64 keeps NumPy's association.
__host__ __device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 acc) {
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y);
    acc.y = fmaf(a.y, b.x, acc.y);
    return acc;
}
// acc + a*b as FOUR CHAINED FMAs in either arithmetic -- the fused pipelines' form.  (The generic cfma above keeps the
// product-then-add association in complex128: mul + fma + add per component, six operations where four do; the operator kernels
// that are held bit-exact to the reference on injected data keep that one.)
template <typename C> __host__ __device__ __forceinline__ C cfma4(C a, C b, C acc) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
    return acc;
}
template <typename C> __host__ __device__ __forceinline__ C cconj(C a) {
    a.y = -a.y;
    return a;
}
template <typename C, typename T> __host__ __device__ __forceinline__ C cscale(C a, T s) {
    a.x *= s;
    a.y *= s;
    return a;
}
// 1 / d and (sqrt(a), 1 / sqrt(a)) for the complex128 SYMBOL LOOPS of the fused pipelines: the hardware estimates v_rcp_f64 /
// v_rsq_f64 + Newton steps -- full precision to a rounding in 5 / 10 instructions, where the IEEE sequences the compiler emits for
// 1.0 / d and sqrt(a) are 11 and ~15 (v_div_scale x 2, v_rcp, five FMAs, v_div_fmas, v_div_fixup).  Not for d = 0 / inf / subnormal
// (the callers' pivots and channel gains are guarded or cannot be).
#ifdef __HIPCC__
__device__ __forceinline__ double rcp_newton(double d) {
    double inv = __builtin_amdgcn_rcp(d);
    inv = fma(fma(-d, inv, 1.0), inv, inv);
    return fma(fma(-d, inv, 1.0), inv, inv);
}
__device__ __forceinline__ void sqrt_rsqrt_newton(double a, double& root, double& inv) {
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    root = g;
    inv = h + h;
}
#endif
template <typename C> __host__ __device__ __forceinline__ C cdivide(C a, C b) {
    auto d = b.x * b.x + b.y * b.y;
    C r;
    r.x = (a.x * b.x + a.y * b.y) / d;
    r.y = (a.y * b.x - a.x * b.y) / d;
    return r;
}

// ---- wave64 reductions ---------------------------------------------------------------------
// Hides a value's provenance from the optimiser.  Used on the thread index at the top of a phase so that the
// LDS addresses derived from it are recomputed there (a few integer ops) instead of being hoisted out of the
// realization loop and kept alive across every other phase, where they spill.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Sum of `len` consecutive complex samples by one wavefront (every lane gets the total).  The run is taken 1280 samples
// at a time (an OFDM symbol of 1024 + cyclic prefix in one trip) with all twenty loads of a lane issued before the first
// add: a loop that waits for each load (or each group of four) spends one memory latency per trip -- under load several
// microseconds -- and that latency, not bandwidth, bounded the tap-mean kernels.
template <typename C>
__device__ __forceinline__ C wave_sum_run(const C* __restrict__ src, int len, int lane) {
    constexpr int kLoads = 20;
    auto re0 = src[0].x * 0, im0 = re0, re1 = re0, im1 = re0, re2 = re0, im2 = re0, re3 = re0, im3 = re0;
    for (int base = 0; base < len; base += 64 * kLoads) {
        C v[kLoads];
#pragma unroll
        for (int u = 0; u < kLoads; ++u) {
            const int j = base + 64 * u + lane;
            v[u].x = v[u].y = re0 * 0;
            if (j < len) v[u] = src[j];
        }
#pragma unroll
        for (int u = 0; u < kLoads; u += 4) {
            re0 += v[u].x; im0 += v[u].y;
            re1 += v[u + 1].x; im1 += v[u + 1].y;
            re2 += v[u + 2].x; im2 += v[u + 2].y;
            re3 += v[u + 3].x; im3 += v[u + 3].y;
        }
    }
    auto re = (re0 + re1) + (re2 + re3), im = (im0 + im1) + (im2 + im3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        re += __shfl_xor(re, off, 64);
        im += __shfl_xor(im, off, 64);
    }
    C r;
    r.x = re;
    r.y = im;
    return r;
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- context --------------------------------------------------------------------------------
struct TwiddleKey {
    int n;
    int dtype;
    bool operator<(const TwiddleKey& o) const { return n != o.n ? n < o.n : dtype < o.dtype; }
};

}  // namespace mcle

struct mcle_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    int n_cu = 0;
    int lds_bytes = 0;
    std::string name;
    // constellation
    int M = 0;
    int bits = 0;
    int kind = MCLE_CONST_GENERIC;
    float2* d_table_f32 = nullptr;
    double2* d_table_f64 = nullptr;
    double qam_scale = 0.0;  // sqrt(2(M-1)/3) for square QAM
    int qam_L = 0;
    // four points, one per quadrant, mirror images of one another in both axes (QPSK = PSK(4) with its pi / 4 offset): the
    // min-distance regions are the quadrants -- modem.hpp: demod_quad_cert.  quad_lut: label of quadrant (re < 0) | (im < 0) << 1
    int quad_ok = 0;
    unsigned quad_lut = 0;
    double quad_min = 0.0, quad_max = 0.0;   // min / max of the points' |re|, |im|
    // four points ON the axes, (+-a, 0) and (0, +-a) -- the reference's PSK(4): exp(j 2 pi m / 4), modulators/fundamental.py:396-448 --
    // whose min-distance regions are bounded by the diagonals: modem.hpp demod_axis4_cert (the complex128 symbol walks, round 6).
    // axis_lut: label of (re - im < 0) | (re + im < 0) << 1, a byte each
    int axis_ok = 0;
    unsigned axis_lut = 0;
    double axis_a = 0.0;
    // M-PSK, M in {8, 16}: M points of one radius at angles 2 pi k / M + phi0 (any label order): the min-distance regions are
    // the M sectors -- modem.hpp: demod_psk_cert.  psk_lut: label of sector k in 64 / M bits each; psk_rot = e^{-j phi0}
    int psk_ok = 0;
    unsigned psk_lut[2] = {};
    double psk_rot[2] = {1.0, 0.0}, psk_radius = 0.0;
    void* d_psk = nullptr;                   // the certificate's constants on the device (modem.hpp: mcle::PskCert)
    // candidate grid of the pruned f32 min-distance search (modem.hpp: DemodGrid); grid_G == 0: none
    unsigned long long* d_grid = nullptr;
    int grid_G = 0;
    float grid_x0 = 0.f, grid_y0 = 0.f, grid_inv_h = 0.f;
    // twiddle tables w[k] = exp(-2 pi i k / n), k < n, per (n, dtype)
    std::map<mcle::TwiddleKey, void*> twiddles;
    // scratch for host->device parameter blocks
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;     // mcle_timer_start / mcle_timer_stop_ms
    hipEvent_t ev_probe0 = nullptr, ev_probe1 = nullptr;   // mcle_hbm_stream_rate's own pair (a probe between the timer calls must not move them)
    // RCCL communicator of the realization-sharded runs (comm.hip); null: single rank
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    void* d_comm_buf = nullptr;
    size_t comm_buf_words = 0;

    // kernel-selection options (mcle_ctx_set_option); 0 = default
    long long opt[MCLE_OPT_COUNT] = {};

    int bind() const;
    int get_twiddles(int n, int dtype, void** d_tw);
    int scratch(size_t bytes, void** d_ptr);
    // a record buffer of `want` bytes that may be SMALLER on a crowded device: halves the request on hipErrorOutOfMemory down to
    // `floor_bytes`, *got = what was obtained (the callers cut their realization slices to it)
    int scratch_upto(size_t want, size_t floor_bytes, void** d_ptr, size_t* got);
};

namespace mcle {
// same-wave LDS hand-off between two passes (the wave's own DS traffic executes in order)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

inline int grid_for(const mcle_ctx* ctx, size_t work_items, int block, int blocks_per_cu = 8) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * blocks_per_cu;
    if (need < 1) need = 1;
    return (int)(need < cap ? need : cap);
}
}  // namespace mcle
