// kernels_channel.hip -- Jakes sum-of-sinusoids fading and the time-varying TDL convolution.
// Reference: channels/fading_generators.py:427-523 (time axis, h = L^-1/2 sum_l exp(j(...))),
// channels/fading.py:949-956 (tap = fading * sqrt(power)), :1080-1090 (SISO corrupt_data).
#include <cstdlib>
#include "fft.hpp"
#include "jakes.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kChBlock = 256;

// Up to kParArg doubles of host parameters as a by-value kernel argument (3.5 KiB of the 4 KiB kernarg segment)
constexpr int kParArg = 448;
struct ParArg {
    double v[kParArg];
};
__global__ void k_unpack_params(ParArg pa, double* __restrict__ dst, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = pa.v[threadIdx.x];
}

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

// The same sum on a uniform time axis (t_i = t0 + i dt) for L <= 16 rays, 128 consecutive samples per wavefront step.
// Around the centre c = i0 - 1/2 of a block,
//     e^{j (w_l t_{c +- (i + 1/2)} + psi_l)} = p_l . V_l[i]   resp.   p_l . conj(V_l[i]) ,
//     p_l = e^{j (w_l t_c + psi_l)} ,   V_l[i] = e^{j w_l dt (i + 1/2)} ,   i < 64 ,
// so a block needs ONE phasor per ray (evaluated as in k_jakes, at the half-sample time t_c) and the four real products
// pr Vr, pi Vi, pr Vi, pi Vr of a ray serve the two samples c + (i + 1/2) and c - (i + 1/2) of lane i: four FMAs per ray
// and PAIR of samples instead of one sincos per ray and sample.  V_l[lane] is L complex values in registers, computed once
// per workgroup and stream and shared through LDS; lane (q, l) = (lane / L, lane % L) evaluates the phasor of ray l for run q of a batch of 64 / L
// runs of kSteps blocks and advances it from block to block by R_l = e^{j w_l dt 128}; the walk reads each phasor back
// from LDS with a wave-uniform address (a broadcast read).  complex128, L = 8: 0.10 -> 0.54 of the 8 TB/s write rate.
// Values equal k_jakes' up to the rounding of w_l t (the reference's own noise floor, eps |w t|):
// fading_generators.py:421-470.
// LT: L rounded up to a multiple of four (rays l >= L carry a zero rotation).
template <typename T, int LT>
__global__ __launch_bounds__(kChBlock) void k_jakes_blocks(const double* __restrict__ par, int L, int n_streams,
                                                           double t0, double dt, const double* __restrict__ amp,
                                                           cx<T>* __restrict__ h, size_t n) {
    const double two_pi = 6.283185307179586476925286766559;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    constexpr int kSteps = 4;                              // blocks a phasor is advanced over before it is re-evaluated
    const int Q = 64 / L;                                  // phasors (runs of kSteps blocks) per batch
    const int q_of = lane / L, l_of = lane - q_of * L;
    const size_t n_blocks = (n + 127) / 128, n_batches = (n_blocks + (size_t)(Q * kSteps) - 1) / (size_t)(Q * kSteps);
    const double* w = par;
    const double* psi = par + (size_t)L * n_streams;
    __shared__ cx<T> s_V[LT][64];                           // the rotation table of the stream, shared by the waves
    __shared__ cx<T> s_p[kChBlock / 64][64];                // each wave's phasors of the current step
    for (int s = blockIdx.y; s < n_streams; s += gridDim.y) {
        __syncthreads();
        for (int l = wave; l < LT; l += waves) {
            const double th = (w[(size_t)(l < L ? l : 0) * n_streams + s] * dt) * ((double)lane + 0.5);   // f64: radians, f32: turns
            double sn, cs;
            sincos(sizeof(T) == 8 ? th : two_pi * th, &sn, &cs);
            s_V[l][lane] = l < L ? mk<T>((T)cs, (T)sn) : mk<T>((T)0, (T)0);
        }
        __syncthreads();
        cx<T> V[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) V[l] = s_V[l][lane];
        const double wl = w[(size_t)l_of * n_streams + s], pl = psi[(size_t)l_of * n_streams + s];
        cx<T> R;                                           // e^{j w_l dt 128}: this lane's ray, one block further
        {
            const double th = (wl * dt) * 128.0;
            double sn, cs;
            sincos(sizeof(T) == 8 ? th : two_pi * th, &sn, &cs);
            R = mk<T>((T)cs, (T)sn);
        }
        const T a = (T)amp[s];
        cx<T>* out = h + (size_t)s * n;
        for (size_t bt = (size_t)blockIdx.x * waves + wave; bt < n_batches; bt += (size_t)gridDim.x * waves) {
            const size_t b0 = bt * (size_t)(Q * kSteps);
            cx<T> p = jakes_ray<T>(wl, pl, jakes_time(t0, dt, (double)((b0 + (size_t)q_of * kSteps) * 128 + 64) - 0.5));
            for (int k = 0; k < kSteps; ++k) {
                // the wave's phasors through LDS (the wave's own DS traffic executes in order; v_readlane into SGPR
                // pairs cost a VALU -> SALU hand-over per operand and ran at a third of this)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                s_p[wave][lane] = p;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                for (int q = 0; q < Q; ++q) {
                    const size_t blk = b0 + (size_t)q * kSteps + k;
                    if (blk * 128 >= n) continue;
                    T A = 0, B = 0, C = 0, D = 0;          // sums of pr Vr, pi Vi, pr Vi, pi Vr
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        const cx<T> pp = s_p[wave][(q * L + l) & 63];       // one address for the whole wave: a broadcast read
                        A = fma(pp.x, V[l].x, A);
                        B = fma(pp.y, V[l].y, B);
                        C = fma(pp.x, V[l].y, C);
                        D = fma(pp.y, V[l].x, D);
                    }
                    const size_t up = blk * 128 + 64 + lane, dn = blk * 128 + 63 - lane;
                    if (up < n) out[up] = mk<T>(a * (A - B), a * (C + D));
                    if (dn < n) out[dn] = mk<T>(a * (A + B), a * (D - C));
                }
                const T nx = fma(p.x, R.x, -(p.y * R.y)), ny = fma(p.x, R.y, p.y * R.x);
                p = mk<T>(nx, ny);
            }
        }
    }
}

// complex64 on the matrix cores (uniform time axis, L <= 16): a tile = 16 groups of 16 consecutive samples;
//     h[16 g + i] = amp sum_l e^{j 2 pi w_l dt i} e^{j (2 pi w_l t_{16 g} + psi_l)}
// is a real [16 x 2 LT] matrix (lane's A operands, per wavefront and stream) times the [2 LT x 16] matrix of ray parts
// at the groups' first samples (one v_sin_f32 per part, f64 phase as everywhere): LT / 2 v_mfma_f32_16x16x4_f32 per plane.
// Lane (j, b) holds ray parts 4 s + b of group j and receives samples 4 b .. 4 b + 3 of group j: 32 contiguous bytes per
// lane, 2 KiB per wavefront and tile.  Same construction as k_run_flat_mfma (pipelines.hip).
typedef float f4j __attribute__((ext_vector_type(4)));
template <int LT>
__global__ __launch_bounds__(kChBlock) void k_jakes_mfma(const double* __restrict__ par, int L, int n_streams, double t0,
                                                         double dt, const double* __restrict__ amp,
                                                         float2* __restrict__ h, size_t n) {
    constexpr int KS = LT / 2;
    const double two_pi = 6.283185307179586476925286766559;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6, j = lane & 15, b = lane >> 4;
    const size_t n_tiles = (n + 255) / 256;
    const double* w = par;
    const double* psi = par + (size_t)L * n_streams;
    for (int s = blockIdx.y; s < n_streams; s += gridDim.y) {
        double wl[KS], pq[KS];
        float a_re[KS], a_im[KS];
        const double a = amp[s];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int l = 2 * k + (b >> 1);
            const bool live = l < L;
            wl[k] = live ? w[(size_t)l * n_streams + s] : 0.0;
            pq[k] = (live ? psi[(size_t)l * n_streams + s] : 0.0) + ((b & 1) ? 0.0 : 0.25);
            double sn, cs;
            sincos(two_pi * ((wl[k] * dt) * (double)j), &sn, &cs);
            a_re[k] = live ? (float)(a * ((b & 1) ? -sn : cs)) : 0.f;
            a_im[k] = live ? (float)(a * ((b & 1) ? cs : sn)) : 0.f;
        }
        float2* out = h + (size_t)s * n;
        for (size_t tile = (size_t)blockIdx.x * waves + wave; tile < n_tiles; tile += (size_t)gridDim.x * waves) {
            const size_t g0 = tile * 256 + 16 * (size_t)j;
            const double tt = jakes_time(t0, dt, (double)g0);
            float bv[KS];
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const double x = fma(wl[k], tt, pq[k]);
                bv[k] = __builtin_amdgcn_sinf((float)__builtin_amdgcn_fract(x));
            }
            f4j hre = {0.f, 0.f, 0.f, 0.f}, him = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                hre = __builtin_amdgcn_mfma_f32_16x16x4f32(a_re[k], bv[k], hre, 0, 0, 0);
                him = __builtin_amdgcn_mfma_f32_16x16x4f32(a_im[k], bv[k], him, 0, 0, 0);
            }
            const size_t i = g0 + 4 * (size_t)b;
            if (i + 4 <= n) {
                float4* o = reinterpret_cast<float4*>(out + i);       // 32-byte aligned when n is even; 8-byte always
                if (((size_t)(out + i) & 15) == 0) {
                    o[0] = make_float4(hre[0], him[0], hre[1], him[1]);
                    o[1] = make_float4(hre[2], him[2], hre[3], him[3]);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) out[i + v] = make_float2(hre[v], him[v]);
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (i + v < n) out[i + v] = make_float2(hre[v], him[v]);
            }
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
                                                          uint32_t stream, size_t row_len, T sigma,
                                                          cx<T>* __restrict__ y) {
    const uint64_t rl = blockIdx.y;
    const Rng rng(seed, first + rl);
    const size_t pairs = (row_len + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (size_t)gridDim.x * blockDim.x) {
        cx<T> z0, z1;
        cn_pair<T>(rng, stream, (uint32_t)p, sigma, z0, z1);
        const size_t o = rl * row_len + 2 * p;
        y[o] = x ? cadd(x[o], z0) : z0;                       // x == nullptr: the draws themselves (mcle_randn_c_batch)
        if (2 * p + 1 < row_len) y[o + 1] = x ? cadd(x[o + 1], z1) : z1;
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
            if (nt == 4) {        // the eight loads of a tap issued before its four products (same accumulation order)
#pragma unroll 2
                for (int i = 0; i < n_taps; ++i) {
                    const long long k = (long long)m - dl.d[i];
                    if (k < 0 || (size_t)k >= n) continue;
                    const cx<T>* gi = g + (((size_t)i * nr + r) * 4) * n + k;
                    const cx<T> g0 = gi[0], g1 = gi[n], g2 = gi[2 * n], g3 = gi[3 * n];
                    const cx<T> x0 = x[k], x1 = x[n + k], x2 = x[2 * n + k], x3 = x[3 * n + k];
                    acc = cadd(acc, cmul(g0, x0));
                    acc = cadd(acc, cmul(g1, x1));
                    acc = cadd(acc, cmul(g2, x2));
                    acc = cadd(acc, cmul(g3, x3));
                }
            } else {
                for (int i = 0; i < n_taps; ++i) {
                    const long long k = (long long)m - dl.d[i];
                    if (k < 0 || (size_t)k >= n) continue;
                    for (int t = 0; t < nt; ++t)
                        acc = cadd(acc, cmul(g[(((size_t)i * nr + r) * nt + t) * n + k], x[(size_t)t * n + k]));
                }
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
            const cx<T> tot = wave_sum_run(src, n + cp, lane);
            const T re = tot.x, im = tot.y;
            if (lane == 0) s_mean[q] = mk<T>(re / (T)(n + cp), im / (T)(n + cp));
        }
        __syncthreads();
#pragma unroll 4
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
    if (host.size() <= (size_t)kParArg) {   // small parameter sets travel as a kernel argument: no copy to wait for
        ParArg pa;
        for (size_t i = 0; i < host.size(); ++i) pa.v[i] = host[i];
        hipLaunchKernelGGL(k_unpack_params, dim3(1), dim3(kParArg), 0, ctx->stream, pa, (double*)d_par, (int)host.size());
        MCLE_LAUNCH_CHECK();
    } else {
        MCLE_HIP(hipMemcpyAsync(d_par, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MCLE_HIP(hipStreamSynchronize(ctx->stream));  // `host` goes out of scope
    }
    const double* d_amp = (const double*)d_par + 2 * np;
    const double* d_times = times ? d_amp + n_streams : nullptr;
    const unsigned gy = (unsigned)(n_streams < 64 ? n_streams : 64);
    if (!times && L <= 16 && n_samples >= 1024 && !ctx->opt[MCLE_OPT_JAKES_DIRECT]) {
        // uniform time axis: 64-sample blocks, one phasor per ray and block (k_jakes_blocks / k_jakes_mfma)
        const int lt = (L + 3) / 4;
        const size_t cap = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 4 / gy + 1;
        if (dtype == MCLE_F32 && !ctx->opt[MCLE_OPT_NO_MFMA]) {      // complex64: the matrix-core form
            const size_t n_tiles = (n_samples + 255) / 256;
            size_t gxm = (n_tiles + 4 * 16 - 1) / (4 * 16);
            if (gxm > 2 * cap) gxm = 2 * cap;
            dim3 gridm((unsigned)gxm, gy);
#define MCLE_JAKES_MFMA(LTT)                                                                                          \
    hipLaunchKernelGGL((k_jakes_mfma<LTT>), gridm, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L, n_streams,  \
                       t0, dt, d_amp, (float2*)d_h, n_samples)
            if (lt == 1) MCLE_JAKES_MFMA(4);
            else if (lt == 2) MCLE_JAKES_MFMA(8);
            else if (lt == 3) MCLE_JAKES_MFMA(12);
            else MCLE_JAKES_MFMA(16);
#undef MCLE_JAKES_MFMA
            MCLE_LAUNCH_CHECK();
            return MCLE_OK;
        }
        const size_t per_batch = (size_t)(64 / L) * 4;               // blocks of 128 samples per batch (kSteps = 4)
        const size_t n_batches = ((n_samples + 127) / 128 + per_batch - 1) / per_batch;
        size_t gxb = (n_batches + 3) / 4;                            // a batch per wavefront, up to 4 waves per SIMD
        if (gxb > cap) gxb = cap;
        dim3 gridb((unsigned)gxb, gy);
#define MCLE_JAKES_BLOCKS(TT, CT, LTT)                                                                                 \
    hipLaunchKernelGGL((k_jakes_blocks<TT, LTT>), gridb, dim3(kChBlock), 0, ctx->stream, (const double*)d_par, L,        \
                       n_streams, t0, dt, d_amp, (CT*)d_h, n_samples)
        if (dtype == MCLE_F32) {
            if (lt == 1) MCLE_JAKES_BLOCKS(float, float2, 4);
            else if (lt == 2) MCLE_JAKES_BLOCKS(float, float2, 8);
            else if (lt == 3) MCLE_JAKES_BLOCKS(float, float2, 12);
            else MCLE_JAKES_BLOCKS(float, float2, 16);
        } else {
            if (lt == 1) MCLE_JAKES_BLOCKS(double, double2, 4);
            else if (lt == 2) MCLE_JAKES_BLOCKS(double, double2, 8);
            else if (lt == 3) MCLE_JAKES_BLOCKS(double, double2, 12);
            else MCLE_JAKES_BLOCKS(double, double2, 16);
        }
#undef MCLE_JAKES_BLOCKS
        MCLE_LAUNCH_CHECK();
        return MCLE_OK;
    }
    unsigned gx = (unsigned)grid_for(ctx, n_samples, kChBlock, 4);
    dim3 grid(gx, gy);
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
    if (n_streams <= kParArg) {              // as a kernel argument: nothing to wait for
        ParArg pa;
        for (int i = 0; i < n_streams; ++i) pa.v[i] = stream_amp[i];
        hipLaunchKernelGGL(k_unpack_params, dim3(1), dim3(kParArg), 0, ctx->stream, pa, (double*)d_amp, n_streams);
        MCLE_LAUNCH_CHECK();
    } else {
        MCLE_HIP(hipMemcpyAsync(d_amp, stream_amp, n_streams * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MCLE_HIP(hipStreamSynchronize(ctx->stream));
    }
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

static int awgn_philox_impl(mcle_ctx* ctx, int dtype, const void* d_x, uint64_t seed, uint64_t first, uint64_t count,
                            uint32_t stream, size_t row_len, double noise_var, void* d_y) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(count <= 65535, "at most 65535 realizations per call");
    MCLE_REQUIRE((uint64_t)row_len < (1ull << 33), "row_len must stay below 2^33 samples (32-bit Philox block counter, two samples per block)");
    if (row_len == 0 || count == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    dim3 grid((unsigned)grid_for(ctx, (row_len + 1) / 2, kChBlock, 2), (unsigned)count);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_awgn_philox<float>, grid, dim3(kChBlock), 0, ctx->stream, (const float2*)d_x, seed, first,
                           stream, row_len, (float)std::sqrt(noise_var), (float2*)d_y);
    else
        hipLaunchKernelGGL(k_awgn_philox<double>, grid, dim3(kChBlock), 0, ctx->stream, (const double2*)d_x, seed,
                           first, stream, row_len, std::sqrt(noise_var), (double2*)d_y);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_awgn_philox(mcle_ctx* ctx, int dtype, const void* d_x, uint64_t seed, uint64_t first, uint64_t count,
                     size_t row_len, double noise_var, void* d_y) {
    MCLE_REQUIRE(d_x != nullptr && d_y != nullptr, "null argument");
    return awgn_philox_impl(ctx, dtype, d_x, seed, first, count, mcle::STREAM_NOISE, row_len, noise_var, d_y);
}

int mcle_randn_c_batch(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first, uint64_t count, uint32_t stream,
                       size_t row_len, double variance, void* d_out) {
    MCLE_REQUIRE(d_out != nullptr, "null argument");
    MCLE_REQUIRE(stream <= 3, "stream must be one of the four mcle-philox-v1 streams");
    return awgn_philox_impl(ctx, dtype, nullptr, seed, first, count, stream, row_len, variance, d_out);
}

}  // extern "C"
