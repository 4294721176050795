// kernels_mimo.hip -- Blast encode / receive filter / decode and the flat MIMO channel, batched.
// Reference: mimo/mimo.py:609-660 (encode / decode, Fortran-order (de)interleave), :264-309 and
// :597-607 (filters), apps/mimo/simulate_mimo.py:96-98 (received = H @ X + noise).
#include "mimo.hpp"

namespace mcle {

constexpr int kMimoBlock = 256;

// X[b][a][c] = x[b][c*nt + a] / sqrt(nt)
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_blast_encode(const cx<T>* __restrict__ x, int nt, size_t ns,
                                                             T inv_root_nt, cx<T>* __restrict__ X) {
    const size_t b = blockIdx.y;
    const size_t n = ns * nt;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t a = i / ns, c = i - a * ns;  // output-major: coalesced writes
        X[b * n + i] = cscale(x[b * n + c * nt + a], inv_root_nt);
    }
}

template <typename T, int NT, int NR>
__global__ __launch_bounds__(64) void k_blast_filter(const cx<T>* __restrict__ Hg, double nv,
                                                     cx<T>* __restrict__ Gg, uint32_t* __restrict__ skipped,
                                                     size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        double2 H[NR][NT], G[NT][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const cx<T> v = Hg[(b * NR + r) * NT + a];
                H[r][a] = mk<double>((double)v.x, (double)v.y);
            }
        const bool ok = blast_filter<NT, NR>(H, nv, G);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < NR; ++r) Gg[(b * NT + a) * NR + r] = mk<T>((T)G[a][r].x, (T)G[a][r].y);
        if (skipped) skipped[b] = ok ? 0u : 1u;
    }
}

// est[b][c*nt + a] = sum_r G[b][a][r] Y[b][r][c]
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_blast_decode(const cx<T>* __restrict__ G, const cx<T>* __restrict__ Y,
                                                             int nr, int nt, size_t ns, cx<T>* __restrict__ est) {
    const size_t b = blockIdx.y;
    const cx<T>* Gb = G + b * (size_t)nt * nr;
    const cx<T>* Yb = Y + b * (size_t)nr * ns;
    cx<T>* eb = est + b * (size_t)nt * ns;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ns; c += (size_t)gridDim.x * blockDim.x) {
        for (int a = 0; a < nt; ++a) {
            cx<T> acc = mk<T>(0, 0);
            for (int r = 0; r < nr; ++r) acc = cfma(Gb[a * nr + r], Yb[(size_t)r * ns + c], acc);
            eb[c * nt + a] = acc;
        }
    }
}

// Y[b][r][c] = sum_a H[b][r][a] X[b][a][c] (+ sigma * noise[b][r][c])
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_mimo_channel(const cx<T>* __restrict__ H, const cx<T>* __restrict__ X,
                                                             const cx<T>* __restrict__ nz, T sigma, int nr, int nt,
                                                             size_t ns, cx<T>* __restrict__ Y) {
    const size_t b = blockIdx.y;
    const cx<T>* Hb = H + b * (size_t)nr * nt;
    const cx<T>* Xb = X + b * (size_t)nt * ns;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ns; c += (size_t)gridDim.x * blockDim.x) {
        for (int r = 0; r < nr; ++r) {
            cx<T> acc = mk<T>(0, 0);
            for (int a = 0; a < nt; ++a) acc = cfma(Hb[r * nt + a], Xb[(size_t)a * ns + c], acc);
            const size_t o = (b * nr + r) * ns + c;
            if (nz) {
                const cx<T> w = nz[o];
                acc.x += sigma * w.x;
                acc.y += sigma * w.y;
            }
            Y[o] = acc;
        }
    }
}

template <typename T, int NT, int NR>
int launch_filter(mcle_ctx* ctx, const void* d_H, double nv, void* d_G, uint32_t* d_skipped, size_t batch) {
    hipLaunchKernelGGL((k_blast_filter<T, NT, NR>), dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream,
                       (const cx<T>*)d_H, nv, (cx<T>*)d_G, d_skipped, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

template <typename T>
int dispatch_filter(mcle_ctx* ctx, int nr, int nt, const void* d_H, double nv, void* d_G, uint32_t* d_skipped,
                    size_t batch) {
#define MCLE_F(NR_, NT_) \
    if (nr == NR_ && nt == NT_) return launch_filter<T, NT_, NR_>(ctx, d_H, nv, d_G, d_skipped, batch);
    MCLE_F(1, 1) MCLE_F(2, 1) MCLE_F(2, 2) MCLE_F(3, 1) MCLE_F(3, 2) MCLE_F(3, 3) MCLE_F(4, 1) MCLE_F(4, 2)
    MCLE_F(4, 3) MCLE_F(4, 4)
#undef MCLE_F
    set_error("unsupported antenna configuration nr=%d nt=%d (need 1 <= nt <= nr <= 4)", nr, nt);
    return MCLE_E_INVAL;
}

int check_mimo(const mcle_ctx* ctx, int dtype, int nr, int nt, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(nt >= 1 && nr >= 1 && nt <= 64 && nr <= 64, "bad antenna counts nr=%d nt=%d", nr, nt);
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    return MCLE_OK;
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_blast_encode(mcle_ctx* ctx, int dtype, const void* d_x, int nt, size_t n, void* d_X, size_t batch) {
    int rc = check_mimo(ctx, dtype, 1, nt, batch);
    if (rc) return rc;
    // mimo.py:633-637
    MCLE_REQUIRE(n % (size_t)nt == 0,
                 "Input array number of elements must be a multiple of the number of transmit antennas.");
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n, kMimoBlock, 4), (unsigned)batch);
    const double s = 1.0 / std::sqrt((double)nt);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_blast_encode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_x, nt,
                           n / nt, (float)s, (float2*)d_X);
    else
        hipLaunchKernelGGL(k_blast_encode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_x, nt,
                           n / nt, s, (double2*)d_X);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_blast_filter(mcle_ctx* ctx, int dtype, const void* d_H, int nr, int nt, double noise_var, void* d_G,
                      uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    // mimo.py:553
    MCLE_REQUIRE(noise_var >= 0.0, "Noise variance must be a non-negative value.");
    MCLE_REQUIRE(nt <= nr, "Blast needs at least as many receive as transmit antennas (nr=%d nt=%d)", nr, nt);
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    if (dtype == MCLE_F32) return dispatch_filter<float>(ctx, nr, nt, d_H, noise_var, d_G, d_skipped, batch);
    return dispatch_filter<double>(ctx, nr, nt, d_H, noise_var, d_G, d_skipped, batch);
}

int mcle_blast_decode(mcle_ctx* ctx, int dtype, const void* d_G, const void* d_Y, int nr, int nt, size_t ns,
                      void* d_est, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, batch);
    if (rc) return rc;
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, ns, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_blast_decode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_G,
                           (const float2*)d_Y, nr, nt, ns, (float2*)d_est);
    else
        hipLaunchKernelGGL(k_blast_decode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_G,
                           (const double2*)d_Y, nr, nt, ns, (double2*)d_est);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_mimo_channel(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_X, const void* d_noise,
                      double noise_var, int nr, int nt, size_t ns, void* d_Y, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, batch);
    if (rc) return rc;
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, ns, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mimo_channel<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_H,
                           (const float2*)d_X, (const float2*)d_noise, (float)std::sqrt(noise_var), nr, nt, ns,
                           (float2*)d_Y);
    else
        hipLaunchKernelGGL(k_mimo_channel<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_H,
                           (const double2*)d_X, (const double2*)d_noise, std::sqrt(noise_var), nr, nt, ns,
                           (double2*)d_Y);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
