// kernels_channel.hip -- Jakes sum-of-sinusoids fading and the time-varying TDL convolution.
// Reference: channels/fading_generators.py:427-523 (time axis, h = L^-1/2 sum_l exp(j(...))),
// channels/fading.py:949-956 (tap = fading * sqrt(power)), :1080-1090 (SISO corrupt_data).
#include "fft.hpp"
#include "jakes.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kChBlock = 256;

// d_par: [2*L*n_streams] doubles = {w[l,s]} then {psi[l,s]} (see jakes.hpp for the meaning of w)
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_jakes(const double* __restrict__ par, int L, int n_streams,
                                                    double t0, double dt, const double* __restrict__ times,
                                                    const double* __restrict__ amp, cx<T>* __restrict__ h, size_t n) {
    const double* w = par;
    const double* psi = par + (size_t)L * n_streams;
    for (int s = blockIdx.y; s < n_streams; s += gridDim.y) {
        const T a = (T)amp[s];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const double t = times ? times[i] : jakes_time(t0, dt, (double)i);
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

// Batched Jakes taps with the phases drawn on-chip (mcle-philox-v1 PHASE stream): realization r,
// stream s: phi[l] = 2 pi u(l*S + s), psi[l] = 2 pi u(L*S + l*S + s) -- the (L, *shape, 1) row-major
// draw order of fading_generators.py:421-425.  taps [count][S][n].  One workgroup per (block of `sb`
// streams, realization): sb*L threads set the rays up in parallel (f64 cos / sincos, the expensive part),
// then all threads sweep the sb rows.
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_jakes_philox(uint64_t seed, uint64_t first, int L, int S, int sb,
                                                           double Fd, double t0, double dt,
                                                           const double* __restrict__ amp, cx<T>* __restrict__ taps,
                                                           size_t n) {
    __shared__ double s_w[kChBlock], s_psi[kChBlock];
    __shared__ float2 s_rot[kChBlock];
    const int s0 = blockIdx.y * sb;
    const int ns = (S - s0) < sb ? (S - s0) : sb;          // streams of this workgroup
    const uint64_t rl = blockIdx.z;
    const Rng rng(seed, first + rl);
    if ((int)threadIdx.x < ns * L) {
        const int sl = threadIdx.x / L, l = threadIdx.x % L, s = s0 + sl;
        const double two_pi = 6.283185307179586476925286766559;
        const double phi = two_pi * uniform_at(rng, STREAM_PHASE, (uint64_t)l * S + s);
        const double psi = two_pi * uniform_at(rng, STREAM_PHASE, (uint64_t)L * S + (uint64_t)l * S + s);
        if (sizeof(T) == 8) {
            s_w[threadIdx.x] = two_pi * Fd * cos(phi);
            s_psi[threadIdx.x] = psi;
        } else {
            s_w[threadIdx.x] = Fd * cos(phi);
            s_psi[threadIdx.x] = psi / two_pi;
            double sn, cs;                                   // one-sample rotation of this ray, exact in f64
            sincos(two_pi * Fd * cos(phi) * dt, &sn, &cs);
            s_rot[threadIdx.x] = make_float2((float)cs, (float)sn);
        }
    }
    __syncthreads();
    // f32: each thread owns kRun consecutive samples; the ray phasor is evaluated exactly (f64 phase) at the
    // first and advanced by the per-sample rotation for the rest (|drift| < 1e-6 over the run).  f64: kRun = 1.
    constexpr int kRun = sizeof(T) == 4 ? 4 : 1;
    const size_t groups = (n + kRun - 1) / kRun;
    for (size_t w = threadIdx.x; w < (size_t)ns * groups; w += blockDim.x) {
        const int sl = (int)(w / groups);
        const size_t i0 = (w - (size_t)sl * groups) * kRun;
        const T a = (T)amp[s0 + sl];
        cx<T>* out = taps + ((size_t)rl * S + s0 + sl) * n;
        const double t = jakes_time(t0, dt, (double)i0);
        T re[kRun], im[kRun];
#pragma unroll
        for (int k = 0; k < kRun; ++k) re[k] = im[k] = 0;
        for (int l = 0; l < L; ++l) {
            cx<T> e = jakes_ray<T>(s_w[sl * L + l], s_psi[sl * L + l], t);
            re[0] += e.x;
            im[0] += e.y;
            if constexpr (kRun > 1) {
                const float2 rot = s_rot[sl * L + l];
#pragma unroll
                for (int k = 1; k < kRun; ++k) {
                    e = cmul(e, rot);
                    re[k] += e.x;
                    im[k] += e.y;
                }
            }
        }
        if constexpr (kRun == 4) {
            if (i0 + kRun <= n && (n % 2 == 0)) {
                float4* o4 = reinterpret_cast<float4*>(out + i0);     // rows start 16-byte aligned when n is even
                o4[0] = make_float4(a * re[0], a * im[0], a * re[1], a * im[1]);
                o4[1] = make_float4(a * re[2], a * im[2], a * re[3], a * im[3]);
                continue;
            }
        }
#pragma unroll
        for (int k = 0; k < kRun; ++k)
            if (i0 + k < n) out[i0 + k] = mk<T>(a * re[k], a * im[k]);
    }
}

// y[r][i] = x[r][i] + sigma * CN(0,1) sample i of (seed, first + r, NOISE)
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_awgn_philox(const cx<T>* __restrict__ x, uint64_t seed, uint64_t first,
                                                          size_t row_len, T sigma, cx<T>* __restrict__ y) {
    const uint64_t rl = blockIdx.y;
    const Rng rng(seed, first + rl);
    const size_t pairs = (row_len + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (size_t)gridDim.x * blockDim.x) {
        cx<T> z0, z1;
        cn_pair<T>(rng, STREAM_NOISE, (uint32_t)p, sigma, z0, z1);
        const size_t o = rl * row_len + 2 * p;
        y[o] = cadd(x[o], z0);
        if (2 * p + 1 < row_len) y[o + 1] = cadd(x[o + 1], z1);
    }
}

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

// MIMO branch of TdlChannel.corrupt_data (fading.py:1107-1117):
//   y[r][m] = sum_i sum_t g[i][r][t][m - d_i] x[t][m - d_i], taps outer / transmit antennas inner (the
//   reference's accumulation order).  x [nt][n], g [S][nr][nt][n], y [nr][n + dmax].
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_tdl_apply_mimo(const cx<T>* __restrict__ x, const cx<T>* __restrict__ g,
                                                             Delays dl, int n_taps, int nr, int nt,
                                                             cx<T>* __restrict__ y, size_t n, size_t n_out) {
    x += (size_t)blockIdx.z * nt * n;
    g += (size_t)blockIdx.z * n_taps * nr * nt * n;
    y += (size_t)blockIdx.z * nr * n_out;
    for (int r = blockIdx.y; r < nr; r += gridDim.y)
        for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < n_out;
             m += (size_t)gridDim.x * blockDim.x) {
            cx<T> acc = mk<T>(0, 0);
            for (int i = 0; i < n_taps; ++i) {
                const long long k = (long long)m - dl.d[i];
                if (k < 0 || (size_t)k >= n) continue;
                for (int t = 0; t < nt; ++t)
                    acc = cadd(acc, cmul(g[(((size_t)i * nr + r) * nt + t) * n + k], x[(size_t)t * n + k]));
            }
            y[(size_t)r * n_out + m] = acc;
        }
}

// Mean frequency response per OFDM symbol on the used subcarriers, for P parallel links (P = nr*nt):
//   Hm[sym][d][p] = sum_i mean_{j in symbol}(g[i][p][j]) w^(bin(d) d_i)
// (TdlImpulseResponse.get_freq_response, fading.py:513-536, averaged over the symbol's fft+cp samples
// as OfdmOneTapEqualizer does, ofdm.py:545-547).  g [S][P][n_sym*(fft+cp)].
template <typename T>
__global__ __launch_bounds__(kChBlock) void k_mean_freq_response(const cx<T>* __restrict__ g, Delays dl, int n_taps,
                                                                 int P, size_t n_sym, int n, int cp, int num_used,
                                                                 bool natural, int mask,
                                                                 const cx<T>* __restrict__ tw,
                                                                 cx<T>* __restrict__ Hm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* s_mean = reinterpret_cast<cx<T>*>(smem);  // [n_taps][P]
    const size_t total = n_sym * (size_t)(n + cp);
    g += (size_t)blockIdx.y * n_taps * P * total;               // batch item
    Hm += (size_t)blockIdx.y * n_sym * num_used * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (size_t sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        __syncthreads();
        for (int q = wave; q < n_taps * P; q += nwave) {  // one wavefront per (tap, link) mean
            const cx<T>* src = g + (size_t)q * total + sym * (size_t)(n + cp);
            T re = 0, im = 0;
            for (int j = lane; j < n + cp; j += 64) {
                re += src[j].x;
                im += src[j].y;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                re += __shfl_xor(re, off, 64);
                im += __shfl_xor(im, off, 64);
            }
            if (lane == 0) s_mean[q] = mk<T>(re / (T)(n + cp), im / (T)(n + cp));
        }
        __syncthreads();
        for (size_t e = threadIdx.x; e < (size_t)num_used * P; e += blockDim.x) {
            const int d = (int)(e / P), p = (int)(e - (size_t)d * P);
            const int k = natural ? d : ofdm_bin(d, n, num_used);
            cx<T> h = mk<T>(0, 0);
            for (int i = 0; i < n_taps; ++i) h = cfma(s_mean[i * P + p], tw[tw_index(k * dl.d[i], n, mask)], h);
            Hm[(sym * num_used + d) * P + p] = h;
        }
    }
}

}  // namespace mcle

using namespace mcle;

extern "C" {

static int jakes_impl(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L, int n_streams, double Fd,
                      double t0, double dt, const double* times, const double* tap_power, void* d_h,
                      size_t n_samples);

int mcle_jakes_generate(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L, int n_streams,
                        double Fd, double t0, double dt, const double* tap_power, void* d_h, size_t n_samples) {
    return jakes_impl(ctx, dtype, phi, psi, L, n_streams, Fd, t0, dt, nullptr, tap_power, d_h, n_samples);
}

int mcle_jakes_generate_at(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L, int n_streams,
                           double Fd, const double* times, const double* tap_power, void* d_h, size_t n_samples) {
    MCLE_REQUIRE(times != nullptr, "null times");
    return jakes_impl(ctx, dtype, phi, psi, L, n_streams, Fd, 0.0, 0.0, times, tap_power, d_h, n_samples);
}

static int jakes_impl(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L, int n_streams, double Fd,
                      double t0, double dt, const double* times, const double* tap_power, void* d_h,
                      size_t n_samples) {
    MCLE_REQUIRE(ctx != nullptr && phi != nullptr && psi != nullptr, "null argument");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(L >= 1 && L <= 1024 && n_streams >= 1 && n_streams <= 65535, "bad L / n_streams");
    if (n_samples == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t np = (size_t)L * n_streams;
    std::vector<double> host(2 * np + n_streams + (times ? n_samples : 0));
    if (times)
        for (size_t i = 0; i < n_samples; ++i) host[2 * np + n_streams + i] = times[i];
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
    const double* d_times = times ? d_amp + n_streams : nullptr;
    unsigned gx = (unsigned)grid_for(ctx, n_samples, kChBlock, 4);
    dim3 grid(gx, (unsigned)(n_streams < 64 ? n_streams : 64));
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_jakes<float>, grid, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L, n_streams,
                           t0, dt, d_times, d_amp, (float2*)d_h, n_samples);
    else
        hipLaunchKernelGGL(k_jakes<double>, grid, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L, n_streams,
                           t0, dt, d_times, d_amp, (double2*)d_h, n_samples);
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

int mcle_tdl_apply_mimo(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_taps, const int32_t* delays,
                        int n_taps, int nr, int nt, void* d_y, size_t n, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && delays != nullptr, "null argument");
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(n_taps >= 1 && n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    MCLE_REQUIRE(nr >= 1 && nt >= 1 && nr <= 64 && nt <= 64, "bad antenna counts");
    Delays dl;
    int maxd = 0;
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) {
        dl.d[i] = i < n_taps ? delays[i] : 0;
        MCLE_REQUIRE(dl.d[i] >= 0, "negative tap delay");
        if (dl.d[i] > maxd) maxd = dl.d[i];
    }
    if (n == 0 || batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t n_out = n + (size_t)maxd;
    dim3 grid((unsigned)grid_for(ctx, n_out, kChBlock, 4), (unsigned)nr, (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_tdl_apply_mimo<float>, grid, dim3(kChBlock), 0, ctx->stream, (const float2*)d_x,
                           (const float2*)d_taps, dl, n_taps, nr, nt, (float2*)d_y, n, n_out);
    else
        hipLaunchKernelGGL(k_tdl_apply_mimo<double>, grid, dim3(kChBlock), 0, ctx->stream, (const double2*)d_x,
                           (const double2*)d_taps, dl, n_taps, nr, nt, (double2*)d_y, n, n_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_tdl_mean_freq_response(mcle_ctx* ctx, int dtype, const void* d_taps, const int32_t* delays, int n_taps,
                                int n_links, size_t n_sym, int fft_size, int cp_size, int num_used, void* d_H,
                                size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && delays != nullptr, "null argument");
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(n_taps >= 1 && n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    MCLE_REQUIRE(n_links >= 1 && n_links <= 64, "n_links must be in [1, 64]");
    MCLE_REQUIRE(fft_size >= 2 && fft_size <= 4096, "fft_size must be in [2, 4096] (got %d)", fft_size);
    // num_used < 0: all fft_size bins in natural order with groups of -num_used samples averaged (1 = the
    // block-static response of corrupt_data_in_freq_domain, fading.py:1126-1287)
    const bool natural = num_used < 0;
    if (natural) {
        MCLE_REQUIRE(-num_used >= 1, "bad group size");
        cp_size = -num_used - fft_size;   // kernel averages fft+cp samples per group
        num_used = fft_size;
    }
    MCLE_REQUIRE(natural || (cp_size >= 0 && cp_size <= fft_size && num_used >= 2 && num_used <= fft_size &&
                             num_used % 2 == 0),
                 "bad OFDM parameters");
    Delays dl;
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) {
        dl.d[i] = i < n_taps ? delays[i] : 0;
        MCLE_REQUIRE(dl.d[i] >= 0, "negative tap delay");
    }
    if (n_sym == 0 || batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(fft_size, dtype, &tw))) return rc;
    const dim3 grid((unsigned)(n_sym < 4096 ? n_sym : 4096), (unsigned)batch);
    const size_t esz = dtype == MCLE_F32 ? sizeof(float2) : sizeof(double2);
    const size_t lds = (size_t)n_taps * n_links * esz;
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mean_freq_response<float>, grid, dim3(kChBlock), lds, ctx->stream,
                           (const float2*)d_taps, dl, n_taps, n_links, n_sym, fft_size, cp_size, num_used, natural,
                           tw_mask_of(fft_size), (const float2*)tw, (float2*)d_H);
    else
        hipLaunchKernelGGL(k_mean_freq_response<double>, grid, dim3(kChBlock), lds, ctx->stream,
                           (const double2*)d_taps, dl, n_taps, n_links, n_sym, fft_size, cp_size, num_used, natural,
                           tw_mask_of(fft_size), (const double2*)tw, (double2*)d_H);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_jakes_taps_philox(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first, uint64_t count, int L,
                           int n_streams, double Fd, double t0, double dt, const double* stream_amp, void* d_taps,
                           size_t n_samples) {
    MCLE_REQUIRE(ctx != nullptr && stream_amp != nullptr, "null argument");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(L >= 1 && L <= 64, "L must be in [1, 64]");
    MCLE_REQUIRE(n_streams >= 1 && n_streams <= 65535 && count <= 65535, "n_streams / count out of range");
    if (n_samples == 0 || count == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    void* d_amp = nullptr;
    if ((rc = ctx->scratch(n_streams * sizeof(double), &d_amp))) return rc;
    MCLE_HIP(hipMemcpyAsync(d_amp, stream_amp, n_streams * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    int sb = kChBlock / L;                                    // streams per workgroup
    if (sb > 16) sb = 16;
    if (sb > n_streams) sb = n_streams;
    dim3 grid(1u, (unsigned)((n_streams + sb - 1) / sb), (unsigned)count);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_jakes_philox<float>, grid, dim3(kChBlock), 0, ctx->stream, seed, first, L, n_streams, sb,
                           Fd, t0, dt, (const double*)d_amp, (float2*)d_taps, n_samples);
    else
        hipLaunchKernelGGL(k_jakes_philox<double>, grid, dim3(kChBlock), 0, ctx->stream, seed, first, L, n_streams,
                           sb, Fd, t0, dt, (const double*)d_amp, (double2*)d_taps, n_samples);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_awgn_philox(mcle_ctx* ctx, int dtype, const void* d_x, uint64_t seed, uint64_t first, uint64_t count,
                     size_t row_len, double noise_var, void* d_y) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(count <= 65535, "at most 65535 realizations per call");
    if (row_len == 0 || count == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    dim3 grid((unsigned)grid_for(ctx, (row_len + 1) / 2, kChBlock, 2), (unsigned)count);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_awgn_philox<float>, grid, dim3(kChBlock), 0, ctx->stream, (const float2*)d_x, seed, first,
                           row_len, (float)std::sqrt(noise_var), (float2*)d_y);
    else
        hipLaunchKernelGGL(k_awgn_philox<double>, grid, dim3(kChBlock), 0, ctx->stream, (const double2*)d_x, seed,
                           first, row_len, std::sqrt(noise_var), (double2*)d_y);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
