// hf_mfma_ab.hip -- A/B of the frequency-response stage of the frequency-selective MIMO-OFDM link (f1) in complex64:
// H(f)[r][a] = sum_s mean[s][r][a] w^(f d_s) for 1024 bins, 16 entries, 5 taps, as
//   V   the product's form (csrc/mimo_tdl_wave.hpp): a lane owns the bins f0, f0 + 512; per entry and tap ONE complex multiply-add
//       (two v_pk_fma_f32) into the tap's delay class (d even / odd; host-sorted class positions, two straight loops), a butterfly
//       behind the loops;
//   M0  the same contraction on the matrix cores: per 16 values of f0 a [16 x K] x [K x 32] real product per delay class
//       (v_mfma_f32_16x16x4_f32: K = 6 -> 8 for the even delays, 4 for the odd ones: six instructions per 32 bins), results left
//       in the accumulator layout (16 lanes x 4 bins per column);
//   M1  M0 + the hand-over the decode needs: a bin's 16 entries in ONE lane (Gram matrix, Cholesky solve per bin), through LDS.
// The consumer is the same checksum (sum of the 16 entries per bin) in V and M1, the plain sum of the accumulators in M0.
// build: hipcc -O3 --offload-arch=gfx950 scripts/experiments/hf_mfma_ab.hip -o scripts/experiments/bin/hf_mfma_ab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float pk2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 1024, S = 5, NR = 4, NT = 4, NE = NR * NT, ITER = 64;

struct Taps {
    int dly[8];
    int ev[4], od[4], ne, no;   // tap indices by delay parity
};

__device__ __forceinline__ pk2 pk_cfma(pk2 a, pk2 b, pk2 acc) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
__host__ __device__ inline float mean_value(unsigned rl, unsigned idx) {        // a cheap reproducible "tap mean"
    unsigned h = rl * 2654435761u + idx * 40503u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(int)(h & 0xFFFFu) * (1.0f / 32768.0f) - 1.0f;
}
__device__ __forceinline__ void make_means(pk2* s_mean, unsigned rl) {
    __syncthreads();
    if (threadIdx.x < S * NE) s_mean[threadIdx.x] = (pk2){mean_value(rl, 2 * threadIdx.x), mean_value(rl, 2 * threadIdx.x + 1)};
    __syncthreads();
}

// ---- V: packed VALU, delay classes ----
__global__ __launch_bounds__(256, 3) void k_valu(Taps tp, const float2* __restrict__ g_tw, float2* __restrict__ out, float2* dump) {
    __shared__ pk2 s_mean[S * NE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    pk2 acc = {0, 0};
    for (int it = 0; it < ITER; ++it) {
        const unsigned rl = blockIdx.x * ITER + it;
        make_means(s_mean, rl);
        for (int wi = w; wi < N / 128; wi += 4) {
            const int f0 = lane + 64 * wi;
            pk2 We[4], Wo[4];                                                   // the product's class positions: even delays, odd delays
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 te = g_tw[(f0 * tp.dly[tp.ev[k]]) & (N - 1)], to = g_tw[(f0 * tp.dly[tp.od[k]]) & (N - 1)];
                We[k] = k < tp.ne ? (pk2){te.x, te.y} : (pk2){0, 0};
                Wo[k] = k < tp.no ? (pk2){to.x, to.y} : (pk2){0, 0};
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                pk2 u[2][NT];
#pragma unroll
                for (int a = 0; a < NT; ++a) u[0][a] = u[1][a] = (pk2){0, 0};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= tp.ne) break;
#pragma unroll
                    for (int a = 0; a < NT; ++a) u[0][a] = pk_cfma(s_mean[(tp.ev[k] * NR + r) * NT + a], We[k], u[0][a]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= tp.no) break;
#pragma unroll
                    for (int a = 0; a < NT; ++a) u[1][a] = pk_cfma(s_mean[(tp.od[k] * NR + r) * NT + a], Wo[k], u[1][a]);
                }
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const pk2 h0 = u[0][a] + u[1][a], h1 = u[0][a] - u[1][a];
                    acc += h0;
                    acc += h1;
                    if (dump && rl == 0) {
                        dump[f0 * NE + r * NT + a] = make_float2(h0.x, h0.y);
                        dump[(f0 + N / 2) * NE + r * NT + a] = make_float2(h1.x, h1.y);
                    }
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(acc.x, acc.y);
}

// ---- M0 / M1: matrix cores ----
// 16x16x4 maps (scripts/experiments/mfma_probe.hip): A[i][k] <- lane i + 16 k, B[k][j] <- lane j + 16 k,
// D[row][col] -> lane col + 16 (row / 4), register row % 4.
template <bool HANDOVER>
__global__ __launch_bounds__(256, 3) void k_mfma(Taps tp, const float2* __restrict__ g_tw, float2* __restrict__ out, float2* dump) {
    __shared__ pk2 s_mean[S * NE];
    __shared__ float s_T[HANDOVER ? 4 * 64 * 36 : 4];                            // per wavefront: 64 bins x 32 values, row pitch 36
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = lane & 15, kk = lane >> 4;
    // this lane's (tap, component) of the A operand per instruction: even class j = 0, 1; odd class
    int a_dly[3], a_cmp[3], b_tap[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = (j < 2 ? 4 * j : 0) + kk, q = k >> 1;
        const bool has = j < 2 ? q < tp.ne : q < tp.no;
        const int s = has ? (j < 2 ? tp.ev[q & 3] : tp.od[q & 3]) : -1;
        b_tap[j] = s;
        a_dly[j] = has ? tp.dly[s] : 0;
        a_cmp[j] = k & 1;
    }
    float accs = 0;
    pk2 acc = {0, 0};
    float* T = s_T + (HANDOVER ? w * 64 * 36 : 0);
    for (int it = 0; it < ITER; ++it) {
        const unsigned rl = blockIdx.x * ITER + it;
        make_means(s_mean, rl);
        // B operands of the realization: column c = 16 t + row -> entry c / 2, part c & 1
        float B[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int c = 16 * t + row, e = c >> 1, part = c & 1;
                float v = 0;
                if (b_tap[j] >= 0) {
                    const pk2 m = s_mean[b_tap[j] * NE + e];
                    v = part == 0 ? (a_cmp[j] == 0 ? m.x : -m.y) : (a_cmp[j] == 0 ? m.y : m.x);
                }
                B[j][t] = v;
            }
        for (int tile = w; tile < N / 32; tile += 4) {                          // 16 values of f0 = 32 bins
            const int f0 = 16 * tile + row;
            float A[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float2 tw = g_tw[(f0 * a_dly[j]) & (N - 1)];
                A[j] = b_tap[j] >= 0 ? (a_cmp[j] ? tw.y : tw.x) : 0.0f;
            }
            f4 ce[2], co[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ce[t] = (f4){0, 0, 0, 0};
                co[t] = (f4){0, 0, 0, 0};
                ce[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], B[0][t], ce[t], 0, 0, 0);
                ce[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1], B[1][t], ce[t], 0, 0, 0);
                co[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2], B[2][t], co[t], 0, 0, 0);
            }
            const int half = (tile >> 2) & 1;                                   // this wavefront's tiles come in pairs: 32 f0 = 64 bins
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f4 h0 = ce[t] + co[t], h1 = ce[t] - co[t];
                if constexpr (!HANDOVER) {
                    accs += h0[0] + h0[1] + h0[2] + h0[3] + h1[0] + h1[1] + h1[2] + h1[3];
                    if (dump && rl == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int f = 16 * tile + 4 * kk + i, c = 16 * t + row;
                            reinterpret_cast<float*>(dump)[(f * NE + (c >> 1)) * 2 + (c & 1)] = h0[i];
                            reinterpret_cast<float*>(dump)[((f + N / 2) * NE + (c >> 1)) * 2 + (c & 1)] = h1[i];
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int b = 16 * half + 4 * kk + i;                   // bin slot of the pair: H(f0) rows 0..31, H(f0 + 512) rows 32..63
                        T[b * 36 + 16 * t + row] = h0[i];
                        T[(32 + b) * 36 + 16 * t + row] = h1[i];
                    }
                }
            }
            if constexpr (HANDOVER) {
                if (half == 1) {                                                // both tiles of the pair are in: a lane takes one bin
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const f4* rowp = reinterpret_cast<const f4*>(T + lane * 36);
                    const int slot = lane & 31;                                 // slots 0 .. 15: the pair's first tile (tile - 4), 16 .. 31: this one
                    const int f = (slot < 16 ? 16 * (tile - 4) + slot : 16 * tile + slot - 16) + (lane >> 5) * (N / 2);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f4 v = rowp[q];
                        acc += (pk2){v[0], v[1]};
                        acc += (pk2){v[2], v[3]};
                        if (dump && rl == 0) {
                            dump[f * NE + 2 * q] = make_float2(v[0], v[1]);
                            dump[f * NE + 2 * q + 1] = make_float2(v[2], v[3]);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = HANDOVER ? make_float2(acc.x, acc.y) : make_float2(accs, 0);
}

int main() {
    Taps tp{};
    const int d[S] = {0, 1, 2, 3, 4};
    for (int s = 0; s < S; ++s) {
        tp.dly[s] = d[s];
        if (d[s] & 1) tp.od[tp.no++] = s; else tp.ev[tp.ne++] = s;
    }
    std::vector<float2> tw(N);
    for (int k = 0; k < N; ++k) tw[k] = make_float2((float)cos(-2 * M_PI * k / N), (float)sin(-2 * M_PI * k / N));
    const int grid = 256 * 12;
    float2 *g_tw, *out, *dump;
    hipMalloc(&g_tw, N * sizeof(float2));
    hipMalloc(&out, grid * 256 * sizeof(float2));
    hipMalloc(&dump, N * NE * sizeof(float2));
    hipMemcpy(g_tw, tw.data(), N * sizeof(float2), hipMemcpyHostToDevice);
    // reference of realization 0 in double
    std::vector<double> ref(N * NE * 2);
    for (int f = 0; f < N; ++f)
        for (int e = 0; e < NE; ++e) {
            double re = 0, im = 0;
            for (int s = 0; s < S; ++s) {
                const double mr = mean_value(0, 2 * (s * NE + e)), mi = mean_value(0, 2 * (s * NE + e) + 1);
                const float2 t = tw[(f * d[s]) & (N - 1)];
                re += mr * t.x - mi * t.y;
                im += mr * t.y + mi * t.x;
            }
            ref[(f * NE + e) * 2] = re;
            ref[(f * NE + e) * 2 + 1] = im;
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[3] = {"V  packed VALU, delay classes", "M0 matrix cores, accumulator layout", "M1 matrix cores + hand-over to one bin per lane"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 3; ++v) {
            auto launch = [&](float2* dmp) {
                if (v == 0) hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, tp, g_tw, out, dmp);
                if (v == 1) hipLaunchKernelGGL(k_mfma<false>, dim3(grid), dim3(256), 0, 0, tp, g_tw, out, dmp);
                if (v == 2) hipLaunchKernelGGL(k_mfma<true>, dim3(grid), dim3(256), 0, 0, tp, g_tw, out, dmp);
            };
            hipMemset(dump, 0, N * NE * sizeof(float2));
            launch(dump);
            hipDeviceSynchronize();
            std::vector<float2> got(N * NE);
            hipMemcpy(got.data(), dump, N * NE * sizeof(float2), hipMemcpyDeviceToHost);
            double worst = 0;
            for (int i = 0; i < N * NE; ++i) {
                worst = fmax(worst, fabs(got[i].x - ref[2 * i]));
                worst = fmax(worst, fabs(got[i].y - ref[2 * i + 1]));
            }
            launch(nullptr);
            hipEventRecord(e0);
            for (int k = 0; k < 5; ++k) launch(nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            const double rl = (double)grid * ITER;
            printf("%-50s %.3f ms per %.0f realizations = %.3e realizations/s of this stage alone (%.0f cycles per realization and CU at 2.4 GHz), max |err| vs f64 %.2e\n",
                   names[v], ms, rl, rl / ms * 1e3, ms * 1e-3 * 2.4e9 * 256 / rl, worst);
        }
    return 0;
}
