// mfma_valu_overlap.hip -- does an f32-input MFMA stream run beside a VALU stream on the same SIMD (gfx950)?
// 512-thread blocks = 2 waves per SIMD; waves 0-3 issue MFMAs, waves 4-7 issue VALU ops; each mode alone and both together.
// Also one wave issuing k VALU ops after every MFMA (the in-wave interleave the fused kernel relies on).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short bf8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 1: mfma only, 2: valu only, 3: both ; KIND 0: 16x16x4 f32, 1: 4x4x1 f32, 2: 32x32x16 bf16
__global__ __launch_bounds__(512) void k_pair(float* out, int kind, int iters) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const float a = 1.0f + threadIdx.x, b = 2.0f;
        if (kind == 0) {
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
        } else if (kind == 1) {
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
            }
        } else {
            bf8 x = {1, 2, 3, 4, 5, 6, 7, 8};
            f16v d0 = {0}, d1 = {0};
            for (int i = 0; i < iters; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, d1, 0, 0, 0);
            }
            c0[0] = d0[0] + d1[3];
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        if (!(MODE & 2)) return;
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        const unsigned k = 0x9E3779B9u;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_xor_b32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                         "v_xor_b32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        }
        r = (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// one wave per SIMD: after every MFMA (16x16x4 f32), K independent VALU ops
template <int K>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters) {
    f4 c0 = {0, 0, 0, 0}, c1 = c0;
    const float a = 1.0f + threadIdx.x, b = 2.0f;
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const unsigned k = 0x9E3779B9u;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(k));
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(k));
    }
    float r = c0[0] + c1[1];
    for (int i = 0; i < 8; ++i) r += (float)v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F> float timed(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipEventRecord(e0, 0);
    f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 512 * 4);
    const int iters = 20000;
    const char* names[3] = {"mfma_f32_16x16x4", "mfma_f32_4x4x1", "mfma_f32_32x32x16_bf16"};
    for (int kind = 0; kind < 3; ++kind) {
        const float m = timed([&] { hipLaunchKernelGGL(k_pair<1>, dim3(256), dim3(512), 0, 0, out, kind, iters); });
        const float v = timed([&] { hipLaunchKernelGGL(k_pair<2>, dim3(256), dim3(512), 0, 0, out, kind, iters); });
        const float b = timed([&] { hipLaunchKernelGGL(k_pair<3>, dim3(256), dim3(512), 0, 0, out, kind, iters); });
        printf("%-24s alone %.3f ms (%.1f ns/MFMA)   VALU wave alone %.3f ms (%.2f ns/inst)   both on one SIMD %.3f ms  -> %s\n",
               names[kind], m, m * 1e6 / (iters * 4.0), v, v * 1e6 / (iters * 16.0), b,
               b < 0.75 * (m + v) ? "overlap" : "serialised");
    }
    printf("one wave, K VALU ops after each 16x16x4 f32 MFMA (ns per MFMA+K group):\n");
    const float t0 = timed([&] { hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t2 = timed([&] { hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t4 = timed([&] { hipLaunchKernelGGL(k_mix<4>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t6 = timed([&] { hipLaunchKernelGGL(k_mix<6>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t8 = timed([&] { hipLaunchKernelGGL(k_mix<8>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t12 = timed([&] { hipLaunchKernelGGL(k_mix<12>, dim3(256), dim3(256), 0, 0, out, iters); });
    const float t16 = timed([&] { hipLaunchKernelGGL(k_mix<16>, dim3(256), dim3(256), 0, 0, out, iters); });
    printf("  K=0 %.1f  K=2 %.1f  K=4 %.1f  K=6 %.1f  K=8 %.1f  K=12 %.1f  K=16 %.1f\n", t0 * 1e6 / (2.0 * iters),
           t2 * 1e6 / (2.0 * iters), t4 * 1e6 / (2.0 * iters), t6 * 1e6 / (2.0 * iters), t8 * 1e6 / (2.0 * iters),
           t12 * 1e6 / (2.0 * iters), t16 * 1e6 / (2.0 * iters));
    return 0;
}
