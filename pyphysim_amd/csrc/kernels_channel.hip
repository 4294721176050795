// kernels_channel.hip -- Jakes sum-of-sinusoids fading and the time-varying TDL convolution.
// Reference: channels/fading_generators.py:427-523 (time axis, h = L^-1/2 sum_l exp(j(...))),
// channels/fading.py:949-956 (tap = fading * sqrt(power)), :1080-1090 (SISO corrupt_data).
#include "jakes.hpp"

namespace mcle {

constexpr int kChBlock = 256;

// d_par: [2*L*n_streams] doubles = {w[l,s]} then {psi[l,s]} (see jakes.hpp for the meaning of w)
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_jakes(const double* __restrict__ par, int L, int n_streams,
                                                    double t0, double dt, const double* __restrict__ amp,
                                                    cx<T>* __restrict__ h, size_t n) {
    const double* w = par;
    const double* psi = par + (size_t)L * n_streams;
    for (int s = blockIdx.y; s < n_streams; s += gridDim.y) {
        const T a = (T)amp[s];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const double t = jakes_time(t0, dt, (double)i);
            T re = 0, im = 0;
            for (int l = 0; l < L; ++l) {
                const cx<T> e = jakes_ray<T>(w[(size_t)l * n_streams + s], psi[(size_t)l * n_streams + s], t);
                re += e.x;
                im += e.y;
            }
            h[(size_t)s * n + i] = mk<T>(a * re, a * im);
        }
    }
}

struct Delays {
    int32_t d[MCLE_MAX_TAPS];
};

// y[m] = sum_i g_i[m - d_i] x[m - d_i], accumulated in tap order like the reference's `+=` loop
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_tdl_apply(const cx<T>* __restrict__ x, const cx<T>* __restrict__ g,
                                                        Delays dl, int n_taps, cx<T>* __restrict__ y, size_t n,
                                                        size_t n_out) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < n_out; m += (size_t)gridDim.x * blockDim.x) {
        cx<T> acc = mk<T>(0, 0);
        for (int i = 0; i < n_taps; ++i) {
            const long long k = (long long)m - dl.d[i];
            if (k >= 0 && (size_t)k < n) {
                const cx<T> p = cmul(g[(size_t)i * n + k], x[k]);
                acc = cadd(acc, p);
            }
        }
        y[m] = acc;
    }
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_jakes_generate(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L, int n_streams,
                        double Fd, double t0, double dt, const double* tap_power, void* d_h, size_t n_samples) {
    MCLE_REQUIRE(ctx != nullptr && phi != nullptr && psi != nullptr, "null argument");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(L >= 1 && L <= 1024 && n_streams >= 1 && n_streams <= 65535, "bad L / n_streams");
    if (n_samples == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t np = (size_t)L * n_streams;
    std::vector<double> host(2 * np + n_streams);
    for (size_t i = 0; i < np; ++i) {
        host[i] = jakes_w(dtype, Fd, phi[i]);
        host[np + i] = jakes_psi(dtype, psi[i]);
    }
    const double inv_sqrt_L = std::sqrt(1.0 / (double)L);
    for (int s = 0; s < n_streams; ++s)
        host[2 * np + s] = inv_sqrt_L * (tap_power ? std::sqrt(tap_power[s]) : 1.0);
    void* d_par = nullptr;
    if ((rc = ctx->scratch(host.size() * sizeof(double), &d_par))) return rc;
    MCLE_HIP(hipMemcpyAsync(d_par, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));  // `host` goes out of scope
    const double* d_amp = (const double*)d_par + 2 * np;
    unsigned gx = (unsigned)grid_for(ctx, n_samples, kChBlock, 4);
    dim3 grid(gx, (unsigned)(n_streams < 64 ? n_streams : 64));
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_jakes<float>, grid, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L, n_streams,
                           t0, dt, d_amp, (float2*)d_h, n_samples);
    else
        hipLaunchKernelGGL(k_jakes<double>, grid, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L, n_streams,
                           t0, dt, d_amp, (double2*)d_h, n_samples);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_tdl_apply(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_taps, const int32_t* delays, int n_taps,
                   void* d_y, size_t n) {
    MCLE_REQUIRE(ctx != nullptr && delays != nullptr, "null argument");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(n_taps >= 1 && n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    Delays dl;
    int maxd = 0;
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) {
        dl.d[i] = i < n_taps ? delays[i] : 0;
        MCLE_REQUIRE(dl.d[i] >= 0, "negative tap delay");
        if (i > 0 && i < n_taps) MCLE_REQUIRE(dl.d[i] > dl.d[i - 1], "tap delays must be strictly increasing");
        if (dl.d[i] > maxd) maxd = dl.d[i];
    }
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t n_out = n + (size_t)maxd;
    const int grid = grid_for(ctx, n_out, kChBlock);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_tdl_apply<float>, dim3(grid), dim3(kChBlock), 0, ctx->stream, (const float2*)d_x,
                           (const float2*)d_taps, dl, n_taps, (float2*)d_y, n, n_out);
    else
        hipLaunchKernelGGL(k_tdl_apply<double>, dim3(grid), dim3(kChBlock), 0, ctx->stream, (const double2*)d_x,
                           (const double2*)d_taps, dl, n_taps, (double2*)d_y, n, n_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
