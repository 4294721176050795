// kernels_hbm.hip -- what the box's HBM delivers to a streaming kernel, measured by the library itself: the denominator of the
// "fraction of the achievable HBM rate" bench.py and scripts/bench_staged_c4.py quote next to the 8 TB/s specification
// ("measure the achievable copy BW on the box and report both").  Until round 4 that figure was torch's
// copy_ of 1 GiB (4.8 TB/s on the driver's box), which the staged chain itself exceeded (5.06 TB/s of algorithmic bytes): a
// library copy is no ceiling.  /opt/skills/guides/MI355X_MICROARCH.md measures 6.29 TB/s with a v4f copy; these are that kernel
// and its relatives, 16-byte accesses, a grid-stride loop with U independent accesses in flight per thread:
//   MCLE_HBM_COPY  b[i] = a[i]              bytes = 2 n        MCLE_HBM_READ  sum of a[i] (one store per thread)   bytes = n
//   MCLE_HBM_TRIAD c[i] = a[i] + s b[i]     bytes = 3 n        MCLE_HBM_WRITE a[i] = s                              bytes = n
#include "common.hpp"

namespace mcle {

typedef float v4f __attribute__((ext_vector_type(4)));     // a native 16-byte vector: what the non-temporal builtins accept

// NT: non-temporal accesses (the streams are touched once: no point in keeping their lines in L2 / MALL)
template <typename V> __device__ __forceinline__ V ld_stream(const V* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename V> __device__ __forceinline__ void st_stream(V* p, V v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int KIND, int U, bool NT>
__global__ __launch_bounds__(256) void k_hbm_stream(const v4f* __restrict__ a, const v4f* __restrict__ b, v4f* __restrict__ c,
                                                    size_t n, float s) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (KIND != MCLE_HBM_WRITE) va[u] = ld_stream(a + i + u * stride, NT);
            if (KIND == MCLE_HBM_TRIAD) vb[u] = ld_stream(b + i + u * stride, NT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (KIND == MCLE_HBM_COPY) st_stream(c + i + u * stride, va[u], NT);
            if (KIND == MCLE_HBM_TRIAD)
                st_stream(c + i + u * stride,
                          va[u] + s * vb[u], NT);
            if (KIND == MCLE_HBM_WRITE) st_stream(c + i + u * stride, (v4f){s, s, s, s}, NT);
            if (KIND == MCLE_HBM_READ) {
                acc += va[u];
            }
        }
    }
    for (; i < n; i += stride) {
        if (KIND == MCLE_HBM_COPY) c[i] = a[i];
        if (KIND == MCLE_HBM_TRIAD) {
            c[i] = a[i] + s * b[i];
        }
        if (KIND == MCLE_HBM_WRITE) c[i] = (v4f){s, s, s, s};
        if (KIND == MCLE_HBM_READ) {
            acc += a[i];
        }
    }
    if (KIND == MCLE_HBM_READ) c[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;   // (c holds one v4f per thread)
}

template <int KIND, int U, bool NT>
static int hbm_stream_rate(mcle_ctx* ctx, size_t bytes, int reps, int blocks_per_cu, double* gbps) {
    const size_t n = bytes / sizeof(v4f);
    const int grid = ctx->n_cu * blocks_per_cu;
    const size_t per_thread = (size_t)grid * 256 * sizeof(v4f);
    // three arrays of `bytes` (a, b, c) in the context's scratch; READ keeps its per-thread sums at the head of c
    void* base = nullptr;
    int rc;
    if ((rc = ctx->scratch(3 * bytes + per_thread, &base))) return rc;
    v4f* a = (v4f*)base;
    v4f* b = a + n;
    v4f* c = b + n;
    MCLE_HIP(hipMemsetAsync(base, 0, 3 * bytes + per_thread, ctx->stream));
    const double moved = (KIND == MCLE_HBM_COPY ? 2.0 : KIND == MCLE_HBM_TRIAD ? 3.0 : 1.0) * (double)(n * sizeof(v4f));
    for (int pass = 0; pass < 2; ++pass) {                  // pass 0 warms the clocks and the page tables
        if (pass) MCLE_HIP(hipEventRecord(ctx->ev_probe0, ctx->stream));
        for (int r = 0; r < (pass ? reps : 2); ++r) {
            hipLaunchKernelGGL((k_hbm_stream<KIND, U, NT>), dim3(grid), dim3(256), 0, ctx->stream, a, b, c, n, 1.0f);
            MCLE_LAUNCH_CHECK();
        }
    }
    MCLE_HIP(hipEventRecord(ctx->ev_probe1, ctx->stream));
    MCLE_HIP(hipEventSynchronize(ctx->ev_probe1));
    float ms = 0.f;
    MCLE_HIP(hipEventElapsedTime(&ms, ctx->ev_probe0, ctx->ev_probe1));
    *gbps = moved * reps / ((double)ms * 1e-3) / 1e9;
    // the three arrays (3.75 GiB at the default 1 GiB) are not kept for the rest of the run: the next pipeline call
    // sizes its own scratch
    if (ctx->scratch_bytes > ((size_t)256 << 20)) {
        MCLE_HIP(hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    return MCLE_OK;
}

}  // namespace mcle

using namespace mcle;

extern "C" int mcle_hbm_stream_rate(mcle_ctx* ctx, int kind, size_t bytes, int reps, int blocks_per_cu, double* gbps) {
    MCLE_REQUIRE(ctx != nullptr && gbps != nullptr, "null argument");
    const int base = kind & 3, nt = (kind >> 2) & 1, u8 = (kind >> 3) & 1;
    MCLE_REQUIRE(kind >= 0 && kind < 16, "kind: MCLE_HBM_COPY / READ / TRIAD / WRITE, | MCLE_HBM_NONTEMPORAL, | MCLE_HBM_UNROLL8");
    MCLE_REQUIRE(bytes >= (size_t)1 << 20 && bytes <= (size_t)8 << 30 && bytes % 16 == 0, "bytes per array: a multiple of 16 in [1 MiB, 8 GiB]");
    MCLE_REQUIRE(reps >= 1 && reps <= 1000 && blocks_per_cu >= 1 && blocks_per_cu <= 64, "reps in [1, 1000], blocks_per_cu in [1, 64]");
    int rc;
    if ((rc = ctx->bind())) return rc;
#define MCLE_HBM_CASE(K_)                                                                                     \
    if (base == K_) {                                                                                         \
        if (nt) return u8 ? hbm_stream_rate<K_, 8, true>(ctx, bytes, reps, blocks_per_cu, gbps)               \
                          : hbm_stream_rate<K_, 4, true>(ctx, bytes, reps, blocks_per_cu, gbps);              \
        return u8 ? hbm_stream_rate<K_, 8, false>(ctx, bytes, reps, blocks_per_cu, gbps)                      \
                  : hbm_stream_rate<K_, 4, false>(ctx, bytes, reps, blocks_per_cu, gbps);                     \
    }
    MCLE_HBM_CASE(MCLE_HBM_COPY) MCLE_HBM_CASE(MCLE_HBM_READ) MCLE_HBM_CASE(MCLE_HBM_TRIAD) MCLE_HBM_CASE(MCLE_HBM_WRITE)
#undef MCLE_HBM_CASE
    return MCLE_E_INVAL;
}
