// kernels_ofdm_mfma.hip -- OFDM.modulate / OFDM.demodulate (reference modulators/ofdm.py:394-466) for complex64 streams
// and fft_size 1024 on the matrix-core transform of fft16.hpp (16 x 16 x 4, DFT-16 passes as v_mfma_f32_16x16x4_f32).
// The radix-4 LDS kernels of kernels_ofdm.hip spend 0.30-0.31 ms on 64 Mi samples (0.43-0.45 of the 8 TB/s spec): they
// are bound by their five transform stages and barriers, not by HBM.  Here four OFDM symbols share a workgroup pass like
// the four antennas of pipeline_mimo_mfma.hip; global traffic is coalesced on both sides (the time samples cross LDS in a
// natural-order layout padded by 4 floats per 64 so that both the butterfly-order and the sample-order accesses are
// bank-conflict free).
#include <cstdlib>

#include "fft.hpp"
#include "fft16.hpp"

namespace mcle {

constexpr int kOmBlock = 256;
constexpr int kOmTimePlane = 1088;                   // 1024 + 4 * 16 floats
constexpr int kOmTimeSlot = 2 * kOmTimePlane;
constexpr int kOmLdsFloats = 4 * kOmTimeSlot;        // >= 4 * kF16Ant: the time staging aliases the planes
__device__ __forceinline__ int om_time_idx(int m) { return m + 4 * (m >> 6); }

struct OmGeom {      // per-thread constants of the 16 x 16 x 4 transform (see pipeline_mimo_mfma.hip)
    int n2, k1p, m2p, k1m, j1m, par, g, gb, plane_g, p1_ld, p1_st, p2_ld, p2_st, mid_off, m0;
};
__device__ __forceinline__ OmGeom om_geom(int tid) {
    OmGeom q;
    const int lane = tid & 63, w = tid >> 6, j = lane & 15;
    q.g = lane >> 4;
    q.gb = q.g >> 1;
    q.n2 = 16 * w + j;
    q.k1p = 4 * w + (j >> 2);
    q.m2p = j & 3;
    const int kkm = ((lane >> 5) << 1) | (lane & 1);
    q.j1m = (lane >> 1) & 15;
    q.k1m = 4 * w + kkm;
    q.par = lane & 1;
    q.plane_g = (q.g & 1) * kF16Plane;
    q.p1_ld = 64 * q.gb + (q.n2 ^ (q.gb * 36));
    q.p1_st = 256 * q.g + (q.n2 ^ (16 * (q.g & 1)));
    const int p2_base = 64 * q.k1p + (q.m2p | f16_swz(q.k1p));
    q.p2_ld = p2_base ^ (4 * q.gb);
    q.p2_st = p2_base ^ (16 * q.g);
    q.mid_off = 64 * q.k1m + ((4 * q.j1m) ^ f16_swz(q.k1m));
    q.m0 = q.k1m + 16 * q.j1m;
    return q;
}

// in [batch][n_in] complex64 (zero padded to whole OFDM symbols) -> out [batch][n_sym * (1024 + cp)]
__global__ __launch_bounds__(kOmBlock, 4) void k_ofdm_mod_1024_mfma(const float2* __restrict__ in, size_t n_in, int cp,
                                                                 int num_used, int n_sym, size_t n_total_sym, float scale,
                                                                 const float2* __restrict__ g_tw, float2* __restrict__ out) {
    constexpr int N = kF16N;
    __shared__ __attribute__((aligned(16))) float s_d[kOmLdsFloats];
    const int tid = threadIdx.x;
    const OmGeom q = om_geom(tid);
    const Dft16Mats mats = dft16_mats(g_tw, tid & 63);
    float2 tw1a[4], tw2a[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        tw1a[x] = g_tw[((4 * q.g + x) * q.n2) & 1023];
        tw2a[x] = g_tw[(16 * (4 * q.g + x) * q.m2p) & 1023];
        if ((q.k1p & 1) && (q.m2p & 1)) tw2a[x] = make_float2(-tw2a[x].x, -tw2a[x].y);
    }
    const size_t n_pass = (n_total_sym + 3) / 4;
    for (size_t ps = blockIdx.x; ps < n_pass; ps += gridDim.x) {
        __syncthreads();                                     // the previous pass has been written out
        if (num_used != N) {
            for (int p = tid; p < 4 * kF16Ant / 4; p += kOmBlock) reinterpret_cast<f4*>(s_d)[p] = f4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
        // ---- symbols -> bins, stored re<->im swapped (inverse transform by the swap identity) ----
        float2 v[4][4];                                      // every global load of the pass in flight before the first LDS store
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const size_t t = 4 * ps + a;
            const bool live = t < n_total_sym;
            const size_t row = live ? t / (size_t)n_sym : 0, sym = live ? t - row * (size_t)n_sym : 0;
            const float2* src = in + row * n_in + sym * (size_t)num_used;
            const size_t left = (live && n_in > sym * (size_t)num_used) ? n_in - sym * (size_t)num_used : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = tid + kOmBlock * i;
                v[a][i] = (d < num_used && (size_t)d < left) ? src[d] : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = tid + kOmBlock * i;
            if (d < num_used) {
                const int pos = f16_pos(ofdm_bin(d, N, num_used));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    s_d[a * kF16Ant + pos] = v[a][i].y;
                    s_d[a * kF16Ant + kF16Plane + pos] = v[a][i].x;
                }
            }
        }
        __syncthreads();
        dft16_pass4(s_d, q.plane_g, mats, tw1a, [&](int t) { return (q.p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t; },
                    [&](int x) { return (q.p1_st ^ ((x << 2) ^ ((x & 1) << 5))) + 64 * x; }, [&]() {});
        __syncthreads();
        dft16_pass4(s_d, q.plane_g, mats, tw2a, [&](int t) { return q.p2_ld ^ (8 * t); },
                    [&](int x) { return q.p2_st ^ (4 * x); }, [&]() {});
        wave_lds_sync();
        float xr[4][4], xi[4][4];                            // [slot sl: time index c = sl ^ 2 par][symbol of the pass]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f4 R = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + q.mid_off);
            const f4 I = *reinterpret_cast<const f4*>(s_d + a * kF16Ant + kF16Plane + q.mid_off);
            const float t0r = R[0] + R[2], t0i = I[0] + I[2], t1r = R[0] - R[2], t1i = I[0] - I[2];
            const float t2r = R[1] + R[3], t2i = I[1] + I[3];
            const float t3r = I[1] - I[3], t3i = R[3] - R[1];     // (z1 - z3) * (-i)
            xi[0][a] = t0r + t2r; xr[0][a] = t0i + t2i;           // planes hold swap(x)
            xi[1][a] = t1r + t3r; xr[1][a] = t1i + t3i;
            xi[2][a] = t0r - t2r; xr[2][a] = t0i - t2i;
            xi[3][a] = t1r - t3r; xr[3][a] = t1i - t3i;
        }
        __syncthreads();                                     // every wave has its samples in registers: restage
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int m = q.m0 + 256 * (sl ^ (2 * q.par));
                s_d[a * kOmTimeSlot + om_time_idx(m)] = xr[sl][a] * scale;
                s_d[a * kOmTimeSlot + kOmTimePlane + om_time_idx(m)] = xi[sl][a] * scale;
            }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const size_t t = 4 * ps + a;
            if (t < n_total_sym) {
                float2* dst = out + t * (size_t)(N + cp);
                for (int jx = tid; jx < N + cp; jx += kOmBlock) {
                    const int n = jx < cp ? N - cp + jx : jx - cp;
                    const int o = a * kOmTimeSlot + om_time_idx(n);
                    dst[jx] = make_float2(s_d[o], s_d[o + kOmTimePlane]);
                }
            }
        }
    }
}

// in [batch][n_sym * (1024 + cp)] -> out [batch][n_sym * num_used]
__global__ __launch_bounds__(kOmBlock, 4) void k_ofdm_demod_1024_mfma(const float2* __restrict__ in, int cp, int num_used,
                                                                   size_t n_total_sym, float scale,
                                                                   const float2* __restrict__ g_tw,
                                                                   float2* __restrict__ out) {
    constexpr int N = kF16N;
    __shared__ __attribute__((aligned(16))) float s_d[kOmLdsFloats];
    const int tid = threadIdx.x;
    const OmGeom q = om_geom(tid);
    const Dft16Mats mats = dft16_mats(g_tw, tid & 63);
    float2 tw1b[4], tw2b[3];
#pragma unroll
    for (int x = 0; x < 4; ++x) tw1b[x] = g_tw[((4 * (4 * q.g + x) + q.m2p) * q.k1p) & 1023];
#pragma unroll
    for (int m2 = 1; m2 < 4; ++m2) {
        tw2b[m2 - 1] = g_tw[(16 * m2 * q.j1m) & 1023];
        if (q.par && (m2 & 1)) tw2b[m2 - 1] = make_float2(-tw2b[m2 - 1].x, -tw2b[m2 - 1].y);
    }
    const size_t n_pass = (n_total_sym + 3) / 4;
    for (size_t ps = blockIdx.x; ps < n_pass; ps += gridDim.x) {
        __syncthreads();
        {
            float2 v[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const size_t t = 4 * ps + a;
                const float2* src = in + (t < n_total_sym ? t : 0) * (size_t)(N + cp) + cp;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[a][i] = t < n_total_sym ? src[tid + kOmBlock * i] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = a * kOmTimeSlot + om_time_idx(tid + kOmBlock * i);
                    s_d[o] = v[a][i].x;
                    s_d[o + kOmTimePlane] = v[a][i].y;
                }
        }
        __syncthreads();
        float yr[4][4], yi[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int m = q.m0 + 256 * (sl ^ (2 * q.par));
                yr[sl][a] = s_d[a * kOmTimeSlot + om_time_idx(m)];
                yi[sl][a] = s_d[a * kOmTimeSlot + kOmTimePlane + om_time_idx(m)];
            }
        __syncthreads();                                     // staging consumed: the planes take over
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // P3': DFT-4 over the slots, x W64^{m2 j1}
            const float t0r = yr[0][r] + yr[2][r], t0i = yi[0][r] + yi[2][r];
            const float t1r = yr[0][r] - yr[2][r], t1i = yi[0][r] - yi[2][r];
            const float t2r = yr[1][r] + yr[3][r], t2i = yi[1][r] + yi[3][r];
            const float t3r = yi[1][r] - yi[3][r], t3i = yr[3][r] - yr[1][r];
            const float2 v1 = cmul_pk(make_float2(t1r + t3r, t1i + t3i), tw2b[0]);
            const float2 v2 = cmul_pk(make_float2(t0r - t2r, t0i - t2i), tw2b[1]);
            const float2 v3 = cmul_pk(make_float2(t1r - t3r, t1i - t3i), tw2b[2]);
            *reinterpret_cast<f4*>(s_d + r * kF16Ant + q.mid_off) = f4{t0r + t2r, v1.x, v2.x, v3.x};
            *reinterpret_cast<f4*>(s_d + r * kF16Ant + kF16Plane + q.mid_off) = f4{t0i + t2i, v1.y, v2.y, v3.y};
        }
        wave_lds_sync();
        dft16_pass4(s_d, q.plane_g, mats, tw1b, [&](int t) { return q.p2_ld ^ (8 * t); },
                    [&](int x) { return q.p2_st ^ (4 * x); }, [&]() {});
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {                        // P1' -> bins 64 (4g + x) + n2, straight to HBM
            const float* pl = s_d + a * kF16Ant + q.plane_g;
            float b[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] = pl[(q.p1_ld ^ (((2 * t) & 7) << 2)) + 128 * t];
            float2 o[4];
            dft16_mfma(mats, b, o);
            const size_t t = 4 * ps + a;
            if (t < n_total_sym) {
                float2* dst = out + t * (size_t)num_used;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int d = ofdm_data_index(64 * (4 * q.g + x) + q.n2, N, num_used);
                    if (d >= 0) dst[d] = make_float2(o[x].x * scale, o[x].y * scale);
                }
            }
        }
    }
}

// host side: MCLE_OK = launched, MCLE_E_UNSUPPORTED = not this kernel's case (complex64, fft_size 1024)
int ofdm_mod_1024_mfma(mcle_ctx* ctx, const void* d_in, size_t n_in, int cp, int num_used, int n_sym, double scale,
                       void* d_out, size_t batch) {
    if (ctx->opt[MCLE_OPT_NO_MFMA]) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(kF16N, MCLE_F32, &tw))) return rc;
    const size_t total = batch * (size_t)n_sym, passes = (total + 3) / 4;
    const size_t cap = (size_t)ctx->n_cu * 4;
    hipLaunchKernelGGL(k_ofdm_mod_1024_mfma, dim3((unsigned)(passes < cap ? passes : cap)), dim3(kOmBlock), 0, ctx->stream,
                       (const float2*)d_in, n_in, cp, num_used, n_sym, total, (float)scale, (const float2*)tw, (float2*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int ofdm_demod_1024_mfma(mcle_ctx* ctx, const void* d_in, int cp, int num_used, int n_sym, double scale, void* d_out,
                         size_t batch) {
    if (ctx->opt[MCLE_OPT_NO_MFMA]) return MCLE_E_UNSUPPORTED;
    int rc;
    void* tw = nullptr;
    if ((rc = ctx->get_twiddles(kF16N, MCLE_F32, &tw))) return rc;
    const size_t total = batch * (size_t)n_sym, passes = (total + 3) / 4;
    const size_t cap = (size_t)ctx->n_cu * 4;
    hipLaunchKernelGGL(k_ofdm_demod_1024_mfma, dim3((unsigned)(passes < cap ? passes : cap)), dim3(kOmBlock), 0, ctx->stream,
                       (const float2*)d_in, cp, num_used, total, (float)scale, (const float2*)tw, (float2*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // namespace mcle
