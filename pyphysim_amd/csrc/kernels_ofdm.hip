// kernels_ofdm.hip -- OFDM.modulate / demodulate and the one-tap equaliser as batched kernels.
// One workgroup per OFDM symbol; the transform runs in LDS (fft.hpp).  Global traffic is one
// coalesced read and one coalesced write of the sample stream (the digit-reversal scatter /
// gather happens on the LDS side).  Reference: modulators/ofdm.py:188-224 (subcarrier map),
// :370-392 (power scale), :394-466 (modulate / demodulate), :515-552 (equaliser).
#include "fft.hpp"

namespace mcle {

constexpr int kOfdmBlock = 256;

// in [batch][n_in] -> out [batch][n_sym*(N+cp)]
template <typename T, int N>
__global__ __launch_bounds__(kOfdmBlock) void k_ofdm_mod(const cx<T>* __restrict__ in, size_t n_in, int cp,
                                                         int num_used, int n_sym, T scale,
                                                         const cx<T>* __restrict__ tw, cx<T>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* s = reinterpret_cast<cx<T>*>(smem);   // [N] samples (bank-swizzled positions)
    cx<T>* s_tw = s + N;                         // [N] twiddles
    const size_t row = blockIdx.y;
    for (int k = threadIdx.x; k < N; k += kOfdmBlock) s_tw[k] = tw[k];
    for (int sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        for (int p = threadIdx.x; p < N; p += kOfdmBlock) s[p] = mk<T>(0, 0);
        __syncthreads();
        const cx<T>* src = in + row * n_in + (size_t)sym * num_used;
        const size_t left = n_in > (size_t)sym * num_used ? n_in - (size_t)sym * num_used : 0;  // zero padding
        {   // every global load of the symbol in flight before the first LDS store (one latency instead of PER)
            constexpr int PER = (N + kOfdmBlock - 1) / kOfdmBlock;
            cx<T> v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int d = threadIdx.x + kOfdmBlock * i;
                v[i] = (d < num_used && (size_t)d < left) ? src[d] : mk<T>(0, 0);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int d = threadIdx.x + kOfdmBlock * i;
                if (d < num_used && (size_t)d < left) s[lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(d, N, num_used)))] = v[i];
            }
        }
        __syncthreads();
        fft_dit<T, N, true, kOfdmBlock, true>(s, 1, N, s_tw);
        cx<T>* dst = out + (row * n_sym + sym) * (size_t)(N + cp);
        for (int j = threadIdx.x; j < N + cp; j += kOfdmBlock) {
            const int n = j < cp ? N - cp + j : j - cp;
            dst[j] = cscale(s[lds_swz<true>(n)], scale);
        }
        __syncthreads();
    }
}

// in [batch][n_sym*(N+cp)] -> out [batch][n_sym*num_used]
template <typename T, int N>
__global__ __launch_bounds__(kOfdmBlock) void k_ofdm_demod(const cx<T>* __restrict__ in, int cp, int num_used,
                                                           int n_sym, T scale, const cx<T>* __restrict__ tw,
                                                           cx<T>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* s = reinterpret_cast<cx<T>*>(smem);
    cx<T>* s_tw = s + N;
    const size_t row = blockIdx.y;
    for (int k = threadIdx.x; k < N; k += kOfdmBlock) s_tw[k] = tw[k];
    for (int sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        const cx<T>* src = in + (row * n_sym + sym) * (size_t)(N + cp) + cp;
        {
            constexpr int PER = (N + kOfdmBlock - 1) / kOfdmBlock;
            cx<T> v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int n = threadIdx.x + kOfdmBlock * i;
                v[i] = n < N ? src[n] : mk<T>(0, 0);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int n = threadIdx.x + kOfdmBlock * i;
                if (n < N) s[lds_swz<true>(n)] = v[i];
            }
        }
        __syncthreads();
        fft_dif<T, N, false, kOfdmBlock, true>(s, 1, N, s_tw);
        cx<T>* dst = out + (row * n_sym + sym) * (size_t)num_used;
        for (int d = threadIdx.x; d < num_used; d += kOfdmBlock)
            dst[d] = cscale(s[lds_swz<true>(fft_pos_of_index<N>(ofdm_bin(d, N, num_used)))], scale);
        __syncthreads();
    }
}

// ---- any fft_size ---------------------------------------------------------------------------
// np.fft takes any length; the radix-4 kernels above take powers of two.  Every other size runs
// here: N = N1 * N2 (host picks the divisor pair closest to sqrt(N)), two direct-DFT passes in
// LDS, N*(N1+N2) complex FMAs per transform (N^2 only for prime N), natural order in and out:
//     A[n2][k1]   = w_N^(n2 k1) * sum_n1 x[N2 n1 + n2] w_N^(N2 n1 k1)
//     X[k1+N1 k2] =               sum_n2 A[n2][k1]     w_N^(N1 n2 k2)
// `a` holds the input and receives the result, `b` is scratch; both [N].  Twiddle indexes are
// stepped with a conditional subtract, never a division.
template <typename T, bool INV>
__device__ __forceinline__ void dft_any(cx<T>* a, cx<T>* b, int N, int N1, int N2, const cx<T>* tw) {
    for (int idx = threadIdx.x; idx < N; idx += kOfdmBlock) {
        const int n2 = idx / N1, k1 = idx - n2 * N1;
        const int step = N2 * k1;                          // < N
        int t = 0;
        cx<T> acc = mk<T>(0, 0);
        for (int n1 = 0; n1 < N1; ++n1) {
            acc = cfma(a[N2 * n1 + n2], tw_get<T, INV>(tw, t), acc);
            t += step;
            if (t >= N) t -= N;
        }
        b[idx] = cmul(acc, tw_get<T, INV>(tw, n2 * k1));   // n2*k1 < N
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += kOfdmBlock) {
        const int k2 = k / N1, k1 = k - k2 * N1;
        const int step = N1 * k2;                          // < N
        int t = 0;
        cx<T> acc = mk<T>(0, 0);
        for (int n2 = 0; n2 < N2; ++n2) {
            acc = cfma(b[n2 * N1 + k1], tw_get<T, INV>(tw, t), acc);
            t += step;
            if (t >= N) t -= N;
        }
        a[k] = acc;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kOfdmBlock) void k_ofdm_mod_any(const cx<T>* __restrict__ in, size_t n_in, int N,
                                                             int N1, int N2, int cp, int num_used, int n_sym,
                                                             T scale, const cx<T>* __restrict__ tw, int tw_in_lds,
                                                             cx<T>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* a = reinterpret_cast<cx<T>*>(smem);
    cx<T>* b = a + N;
    const cx<T>* w = tw;
    if (tw_in_lds) {
        cx<T>* s_tw = b + N;
        for (int k = threadIdx.x; k < N; k += kOfdmBlock) s_tw[k] = tw[k];
        w = s_tw;
    }
    const size_t row = blockIdx.y;
    for (int sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        for (int p = threadIdx.x; p < N; p += kOfdmBlock) a[p] = mk<T>(0, 0);
        __syncthreads();
        const cx<T>* src = in + row * n_in + (size_t)sym * num_used;
        const size_t left = n_in > (size_t)sym * num_used ? n_in - (size_t)sym * num_used : 0;
        for (int d = threadIdx.x; d < num_used; d += kOfdmBlock)
            if ((size_t)d < left) a[ofdm_bin(d, N, num_used)] = src[d];
        __syncthreads();
        dft_any<T, true>(a, b, N, N1, N2, w);
        cx<T>* dst = out + (row * n_sym + sym) * (size_t)(N + cp);
        for (int j = threadIdx.x; j < N + cp; j += kOfdmBlock) dst[j] = cscale(a[j < cp ? N - cp + j : j - cp], scale);
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(kOfdmBlock) void k_ofdm_demod_any(const cx<T>* __restrict__ in, int N, int N1, int N2,
                                                               int cp, int num_used, int n_sym, T scale,
                                                               const cx<T>* __restrict__ tw, int tw_in_lds,
                                                               cx<T>* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* a = reinterpret_cast<cx<T>*>(smem);
    cx<T>* b = a + N;
    const cx<T>* w = tw;
    if (tw_in_lds) {
        cx<T>* s_tw = b + N;
        for (int k = threadIdx.x; k < N; k += kOfdmBlock) s_tw[k] = tw[k];
        w = s_tw;
    }
    const size_t row = blockIdx.y;
    for (int sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        const cx<T>* src = in + (row * n_sym + sym) * (size_t)(N + cp) + cp;
        for (int n = threadIdx.x; n < N; n += kOfdmBlock) a[n] = src[n];
        __syncthreads();
        dft_any<T, false>(a, b, N, N1, N2, w);
        cx<T>* dst = out + (row * n_sym + sym) * (size_t)num_used;
        for (int d = threadIdx.x; d < num_used; d += kOfdmBlock) dst[d] = cscale(a[ofdm_bin(d, N, num_used)], scale);
        __syncthreads();
    }
}

// One workgroup per OFDM symbol: mean of each sparse tap over the symbol's N+cp samples (CP
// included, ofdm.py:545-547), then H[k] = sum_i mean_i w^(k d_i) (== mean over samples of the
// per-sample FFTs of fading.py:513-536, by linearity), then data / H on the used bins.
struct TapDelays {
    int32_t d[MCLE_MAX_TAPS];
};
template <typename T>
__global__ __launch_bounds__(kOfdmBlock) void k_onetap_eq(const cx<T>* __restrict__ data,
                                                          const cx<T>* __restrict__ taps, TapDelays delays,
                                                          int n_taps, size_t n_sym, int n, int cp, int num_used,
                                                          int mask, const cx<T>* __restrict__ tw,
                                                          cx<T>* __restrict__ out) {
    __shared__ cx<T> s_mean[MCLE_MAX_TAPS];
    const size_t total = n_sym * (size_t)(n + cp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t sym = blockIdx.x; sym < n_sym; sym += gridDim.x) {
        for (int i = wave; i < n_taps; i += kOfdmBlock / 64) {     // one wavefront per tap mean
            const cx<T> tot = wave_sum_run(taps + (size_t)i * total + sym * (size_t)(n + cp), n + cp, lane);
            if (lane == 0) s_mean[i] = mk<T>(tot.x / (T)(n + cp), tot.y / (T)(n + cp));
        }
        __syncthreads();
#pragma unroll 4
        for (int d = threadIdx.x; d < num_used; d += blockDim.x) {      // four bins' loads in flight per thread
            const int k = ofdm_bin(d, n, num_used);
            const size_t o = sym * (size_t)num_used + d;
            const cx<T> y = data[o];
            cx<T> h = mk<T>(0, 0);
            for (int i = 0; i < n_taps; ++i) h = cfma(s_mean[i], tw[tw_index(k * delays.d[i], n, mask)], h);
            out[o] = cdivide(y, h);
        }
        __syncthreads();
    }
}

int ofdm_mod_1024_mfma(mcle_ctx* ctx, const void* d_in, size_t n_in, int cp, int num_used, int n_sym, double scale,
                       void* d_out, size_t batch);          // kernels_ofdm_mfma.hip
int ofdm_demod_1024_mfma(mcle_ctx* ctx, const void* d_in, int cp, int num_used, int n_sym, double scale, void* d_out,
                         size_t batch);

int check_ofdm(const mcle_ctx* ctx, int dtype, int fft_size, int cp_size, int num_used) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(fft_size >= 2 && fft_size <= 4096, "fft_size must be in [2, 4096] (got %d)", fft_size);
    // same messages as OFDM.set_parameters (ofdm.py:75-90)
    MCLE_REQUIRE(cp_size >= 0 && cp_size <= fft_size,
                 "cp_size must be nonnegative and cannot be greater than fft_size");
    MCLE_REQUIRE(num_used <= fft_size, "Number of used subcarriers cannot be greater than the fft_size");
    MCLE_REQUIRE(num_used >= 2 && num_used % 2 == 0, "Number of used subcarriers must be a multiple of 2");
    return MCLE_OK;
}

template <typename T, int N>
int launch_mod(mcle_ctx* ctx, const void* d_in, size_t n_in, int cp, int num_used, int n_sym, double scale,
               const void* tw, void* d_out, size_t batch) {
    const unsigned gx = (unsigned)(n_sym < 4096 ? n_sym : 4096);
    MCLE_HIP(hipFuncSetAttribute((const void*)k_ofdm_mod<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(2 * N * sizeof(cx<T>))));
    hipLaunchKernelGGL((k_ofdm_mod<T, N>), dim3(gx, (unsigned)batch), dim3(kOfdmBlock), 2 * N * sizeof(cx<T>),
                       ctx->stream, (const cx<T>*)d_in, n_in, cp, num_used, n_sym, (T)scale, (const cx<T>*)tw,
                       (cx<T>*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}
template <typename T, int N>
int launch_demod(mcle_ctx* ctx, const void* d_in, int cp, int num_used, int n_sym, double scale, const void* tw,
                 void* d_out, size_t batch) {
    const unsigned gx = (unsigned)(n_sym < 4096 ? n_sym : 4096);
    MCLE_HIP(hipFuncSetAttribute((const void*)k_ofdm_demod<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(2 * N * sizeof(cx<T>))));
    hipLaunchKernelGGL((k_ofdm_demod<T, N>), dim3(gx, (unsigned)batch), dim3(kOfdmBlock), 2 * N * sizeof(cx<T>),
                       ctx->stream, (const cx<T>*)d_in, cp, num_used, n_sym, (T)scale, (const cx<T>*)tw,
                       (cx<T>*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

// any-size path: LDS = a[N] + b[N] (+ twiddles when the three fit comfortably)
template <typename T> size_t any_lds(int n, int* tw_in_lds) {
    *tw_in_lds = 3 * (size_t)n * sizeof(cx<T>) <= 96 * 1024;
    return (size_t)(*tw_in_lds ? 3 : 2) * n * sizeof(cx<T>);
}
template <typename T>
int launch_mod_any(mcle_ctx* ctx, const void* d_in, size_t n_in, int n, int cp, int num_used, int n_sym,
                   double scale, const void* tw, void* d_out, size_t batch) {
    int n1, n2, in_lds;
    dft_any_split(n, &n1, &n2);
    const size_t lds = any_lds<T>(n, &in_lds);
    const unsigned gx = (unsigned)(n_sym < 4096 ? n_sym : 4096);
    MCLE_HIP(hipFuncSetAttribute((const void*)k_ofdm_mod_any<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL((k_ofdm_mod_any<T>), dim3(gx, (unsigned)batch), dim3(kOfdmBlock), lds, ctx->stream,
                       (const cx<T>*)d_in, n_in, n, n1, n2, cp, num_used, n_sym, (T)scale, (const cx<T>*)tw, in_lds,
                       (cx<T>*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}
template <typename T>
int launch_demod_any(mcle_ctx* ctx, const void* d_in, int n, int cp, int num_used, int n_sym, double scale,
                     const void* tw, void* d_out, size_t batch) {
    int n1, n2, in_lds;
    dft_any_split(n, &n1, &n2);
    const size_t lds = any_lds<T>(n, &in_lds);
    const unsigned gx = (unsigned)(n_sym < 4096 ? n_sym : 4096);
    MCLE_HIP(hipFuncSetAttribute((const void*)k_ofdm_demod_any<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL((k_ofdm_demod_any<T>), dim3(gx, (unsigned)batch), dim3(kOfdmBlock), lds, ctx->stream,
                       (const cx<T>*)d_in, n, n1, n2, cp, num_used, n_sym, (T)scale, (const cx<T>*)tw, in_lds,
                       (cx<T>*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

#define MCLE_FFT_SWITCH(N_, CALL)                              \
    switch (N_) {                                              \
        case 16: { constexpr int NN = 16; CALL; } break;       \
        case 32: { constexpr int NN = 32; CALL; } break;       \
        case 64: { constexpr int NN = 64; CALL; } break;       \
        case 128: { constexpr int NN = 128; CALL; } break;     \
        case 256: { constexpr int NN = 256; CALL; } break;     \
        case 512: { constexpr int NN = 512; CALL; } break;     \
        case 1024: { constexpr int NN = 1024; CALL; } break;   \
        case 2048: { constexpr int NN = 2048; CALL; } break;   \
        case 4096: { constexpr int NN = 4096; CALL; } break;   \
        default: set_error("unsupported fft size %d", N_); rc = MCLE_E_INVAL; \
    }

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_ofdm_modulate(mcle_ctx* ctx, int dtype, const void* d_in, size_t n_in, int fft_size, int cp_size,
                       int num_used, void* d_out, size_t batch) {
    int rc = check_ofdm(ctx, dtype, fft_size, cp_size, num_used);
    if (rc) return rc;
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    if (n_in == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(fft_size, dtype, &tw))) return rc;
    const int n_sym = (int)((n_in + num_used - 1) / num_used);
    // sqrt(fft^2/(used+cp)) * (1/fft of numpy's ifft)   (ofdm.py:390-391,421-422)
    const double scale = std::sqrt((double)fft_size * fft_size / ((double)num_used + cp_size)) / fft_size;
    if (!fft_is_radix4_size(fft_size))
        return dtype == MCLE_F32
                   ? launch_mod_any<float>(ctx, d_in, n_in, fft_size, cp_size, num_used, n_sym, scale, tw, d_out, batch)
                   : launch_mod_any<double>(ctx, d_in, n_in, fft_size, cp_size, num_used, n_sym, scale, tw, d_out,
                                            batch);
    if (dtype == MCLE_F32 && fft_size == 1024) {   // complex64, 1024 points: the matrix-core transform (kernels_ofdm_mfma.hip)
        rc = ofdm_mod_1024_mfma(ctx, d_in, n_in, cp_size, num_used, n_sym, scale, d_out, batch);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    if (dtype == MCLE_F32) {
        MCLE_FFT_SWITCH(fft_size, rc = (launch_mod<float, NN>(ctx, d_in, n_in, cp_size, num_used, n_sym, scale, tw,
                                                               d_out, batch)));
    } else {
        MCLE_FFT_SWITCH(fft_size, rc = (launch_mod<double, NN>(ctx, d_in, n_in, cp_size, num_used, n_sym, scale, tw,
                                                                d_out, batch)));
    }
    return rc;
}

int mcle_ofdm_demodulate(mcle_ctx* ctx, int dtype, const void* d_in, size_t n_sym, int fft_size, int cp_size,
                         int num_used, void* d_out, size_t batch) {
    int rc = check_ofdm(ctx, dtype, fft_size, cp_size, num_used);
    if (rc) return rc;
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    if (n_sym == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(fft_size, dtype, &tw))) return rc;
    const double scale = 1.0 / std::sqrt((double)fft_size * fft_size / ((double)num_used + cp_size));
    if (!fft_is_radix4_size(fft_size))
        return dtype == MCLE_F32 ? launch_demod_any<float>(ctx, d_in, fft_size, cp_size, num_used, (int)n_sym, scale,
                                                           tw, d_out, batch)
                                 : launch_demod_any<double>(ctx, d_in, fft_size, cp_size, num_used, (int)n_sym, scale,
                                                            tw, d_out, batch);
    if (dtype == MCLE_F32 && fft_size == 1024) {
        rc = ofdm_demod_1024_mfma(ctx, d_in, cp_size, num_used, (int)n_sym, scale, d_out, batch);
        if (rc != MCLE_E_UNSUPPORTED) return rc;
    }
    if (dtype == MCLE_F32) {
        MCLE_FFT_SWITCH(fft_size, rc = (launch_demod<float, NN>(ctx, d_in, cp_size, num_used, (int)n_sym, scale, tw,
                                                                 d_out, batch)));
    } else {
        MCLE_FFT_SWITCH(fft_size, rc = (launch_demod<double, NN>(ctx, d_in, cp_size, num_used, (int)n_sym, scale,
                                                                  tw, d_out, batch)));
    }
    return rc;
}

int mcle_onetap_equalize(mcle_ctx* ctx, int dtype, const void* d_data, const void* d_taps, const int32_t* delays,
                         int n_taps, size_t n_sym, int fft_size, int cp_size, int num_used, void* d_out) {
    int rc = check_ofdm(ctx, dtype, fft_size, cp_size, num_used);
    if (rc) return rc;
    MCLE_REQUIRE(delays != nullptr, "null delays");
    MCLE_REQUIRE(n_taps >= 1 && n_taps <= MCLE_MAX_TAPS, "n_taps must be in [1, %d]", MCLE_MAX_TAPS);
    if (n_sym == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(fft_size, dtype, &tw))) return rc;
    TapDelays td;
    for (int i = 0; i < MCLE_MAX_TAPS; ++i) td.d[i] = i < n_taps ? delays[i] : 0;
    for (int i = 0; i < n_taps; ++i) MCLE_REQUIRE(td.d[i] >= 0, "negative tap delay");
    const unsigned grid = (unsigned)(n_sym < 8192 ? n_sym : 8192);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_onetap_eq<float>, dim3(grid), dim3(kOfdmBlock), 0, ctx->stream, (const float2*)d_data,
                           (const float2*)d_taps, td, n_taps, n_sym, fft_size, cp_size, num_used, tw_mask_of(fft_size),
                           (const float2*)tw, (float2*)d_out);
    else
        hipLaunchKernelGGL(k_onetap_eq<double>, dim3(grid), dim3(kOfdmBlock), 0, ctx->stream, (const double2*)d_data,
                           (const double2*)d_taps, td, n_taps, n_sym, fft_size, cp_size, num_used,
                           tw_mask_of(fft_size), (const double2*)tw, (double2*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
