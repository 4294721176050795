// f32_rates.hip -- issue cost of the VALU ops the complex64 pipelines are made of (cycles per wave-instruction per SIMD at
// 1 / 2 / 4 wavefronts per SIMD, 2.4 GHz assumed): plain and packed f32 arithmetic, the Philox integer ops, transcendentals.
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 scripts/experiments/f32_rates.hip -o /tmp/f32_rates && /tmp/f32_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
// eight independent registers a0..a7 (scalars) or p0..p7 (pairs), 1024 x 8 x 8 instructions per thread
#define BODY32(NAME, ASM)                                                                                              \
    __global__ void NAME(float* out) {                                                                                 \
        float a0 = 1.f + threadIdx.x * 1e-3f, a1 = a0 + 1e-3f, a2 = a0 + 2e-3f, a3 = a0 + 3e-3f, a4 = a0 + 4e-3f,      \
              a5 = a0 + 5e-3f, a6 = a0 + 6e-3f, a7 = a0 + 7e-3f, k = 1.0000001f;                                       \
        for (int i = 0; i < 1024; ++i) { REP8(ASM) }                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + k;                        \
    }
#define BODYPK(NAME, ASM)                                                                                              \
    __global__ void NAME(float* out) {                                                                                 \
        v2f a0 = {1.f + threadIdx.x * 1e-3f, 0.5f}, a1 = a0 + 1e-3f, a2 = a0 + 2e-3f, a3 = a0 + 3e-3f, a4 = a0 + 4e-3f, \
            a5 = a0 + 5e-3f, a6 = a0 + 6e-3f, a7 = a0 + 7e-3f, k = {1.0000001f, 0.9999999f};                           \
        for (int i = 0; i < 1024; ++i) { REP8(ASM) }                                                                   \
        const v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + k;                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                                        \
    }
#define BODYU(NAME, ASM)                                                                                               \
    __global__ void NAME(float* out) {                                                                                 \
        unsigned a0 = threadIdx.x * 2654435761u + 1u, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 9u,          \
                 a5 = a0 * 11u, a6 = a0 * 13u, a7 = a0 * 15u, k = 0xD2511F53u;                                         \
        for (int i = 0; i < 1024; ++i) { REP8(ASM) }                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ k);               \
    }
#define IO8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)
#define A8(OP, MOD) asm volatile(OP " %0, %0, %8" MOD "\n" OP " %1, %1, %8" MOD "\n" OP " %2, %2, %8" MOD "\n" OP " %3, %3, %8" MOD "\n" \
                                 OP " %4, %4, %8" MOD "\n" OP " %5, %5, %8" MOD "\n" OP " %6, %6, %8" MOD "\n" OP " %7, %7, %8" MOD "\n" IO8);
#define F8(OP, MOD) asm volatile(OP " %0, %0, %8, %0" MOD "\n" OP " %1, %1, %8, %1" MOD "\n" OP " %2, %2, %8, %2" MOD "\n" OP " %3, %3, %8, %3" MOD "\n" \
                                 OP " %4, %4, %8, %4" MOD "\n" OP " %5, %5, %8, %5" MOD "\n" OP " %6, %6, %8, %6" MOD "\n" OP " %7, %7, %8, %7" MOD "\n" IO8);
#define U8(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n" IO8);
#define DEPF(OP, MOD) asm volatile(OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" \
                                   OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" OP " %0, %0, %8, %0" MOD "\n" IO8);

BODY32(k_add_f32, A8("v_add_f32", ""))
BODY32(k_mul_f32, A8("v_mul_f32", ""))
BODY32(k_fma_f32, F8("v_fma_f32", ""))
BODY32(k_fma_f32_dep, DEPF("v_fma_f32", ""))
BODY32(k_log_f32, U8("v_log_f32"))
BODY32(k_sin_f32, U8("v_sin_f32"))
BODY32(k_sqrt_f32, U8("v_sqrt_f32"))
BODY32(k_rcp_f32, U8("v_rcp_f32"))
BODY32(k_rndne_f32, U8("v_rndne_f32"))
BODY32(k_max_f32, A8("v_max_f32", ""))
BODYPK(k_pk_add_f32, A8("v_pk_add_f32", ""))
BODYPK(k_pk_add_f32_swz, A8("v_pk_add_f32", " op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"))
BODYPK(k_pk_mul_f32, A8("v_pk_mul_f32", ""))
BODYPK(k_pk_mul_f32_swz, A8("v_pk_mul_f32", " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"))
BODYPK(k_pk_fma_f32, F8("v_pk_fma_f32", ""))
BODYPK(k_pk_fma_f32_swz, F8("v_pk_fma_f32", " op_sel:[0,0,0] op_sel_hi:[0,1,1]"))
BODYPK(k_pk_fma_f32_dep, DEPF("v_pk_fma_f32", ""))
BODYPK(k_pk_mov_b32, A8("v_pk_mov_b32", " op_sel:[1,0]"))
BODYU(k_mul_lo_u32, A8("v_mul_lo_u32", ""))
BODYU(k_mul_hi_u32, A8("v_mul_hi_u32", ""))
BODYU(k_xor_b32, A8("v_xor_b32", ""))
BODYU(k_bitop3, F8("v_bitop3_b32", " bitop3:0x96"))
BODYU(k_add3_u32, F8("v_add3_u32", ""))
BODYU(k_lshl_add_u32, F8("v_lshl_add_u32", ""))
BODYU(k_cndmask, A8("v_cndmask_b32", ", vcc"))
BODYU(k_mul_u32_u24, A8("v_mul_u32_u24", ""))
BODYU(k_mad_u32_u24, F8("v_mad_u32_u24", ""))
// v_cndmask_b32 reading its mask from vcc / from an SGPR pair, back to back and behind the v_cmp that writes the mask
BODYU(k_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\n"
                                   "v_cndmask_b32 %4, %4, %8, s[20:21]\nv_cndmask_b32 %5, %5, %8, s[20:21]\nv_cndmask_b32 %6, %6, %8, s[20:21]\nv_cndmask_b32 %7, %7, %8, s[20:21]\n" IO8 : "s20", "s21");)
BODYU(k_cmp_cndmask, asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                                  "v_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n" IO8 : "vcc");)
BODYU(k_cmp_u32, asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cmp_lt_u32 vcc, %1, %8\nv_cmp_lt_u32 vcc, %2, %8\nv_cmp_lt_u32 vcc, %3, %8\n"
                              "v_cmp_lt_u32 vcc, %4, %8\nv_cmp_lt_u32 vcc, %5, %8\nv_cmp_lt_u32 vcc, %6, %8\nv_cmp_lt_u32 vcc, %7, %8\n" IO8 : "vcc");)
BODYU(k_cndmask_xor, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_xor_b32 %1, %1, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_xor_b32 %3, %3, %8\n"
                                  "v_cndmask_b32 %4, %4, %8, vcc\nv_xor_b32 %5, %5, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_xor_b32 %7, %7, %8\n" IO8);)
BODYU(k_cndmask_dst, asm volatile("v_cndmask_b32 %0, %1, %8, vcc\nv_cndmask_b32 %1, %2, %8, vcc\nv_cndmask_b32 %2, %3, %8, vcc\nv_cndmask_b32 %3, %4, %8, vcc\n"
                                  "v_cndmask_b32 %4, %5, %8, vcc\nv_cndmask_b32 %5, %6, %8, vcc\nv_cndmask_b32 %6, %7, %8, vcc\nv_cndmask_b32 %7, %0, %8, vcc\n" IO8);)
BODY32(k_cvt_f32_u32, U8("v_cvt_f32_u32"))
BODY32(k_cvt_i32_f32, U8("v_cvt_i32_f32"))
BODY32(k_fmac_f32, A8("v_fmac_f32", ""))
BODY32(k_sub_f32, A8("v_sub_f32", ""))
BODY32(k_cos_f32, U8("v_cos_f32"))
BODYU(k_lshrrev, A8("v_lshrrev_b32", ""))
BODYU(k_and_b32, A8("v_and_b32", ""))
BODYU(k_add_u32, A8("v_add_u32", ""))
BODYU(k_bfe_u32, F8("v_bfe_u32", ""))
BODYU(k_alignbit, F8("v_alignbit_b32", ""))
BODYU(k_perm, F8("v_perm_b32", ""))
// v_mad_u64_u32: 64-bit destination + carry-out pair; four independent destinations
__global__ void k_mad_u64_u32(float* out) {
    unsigned a0 = threadIdx.x * 2654435761u + 1u, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, k = 0xD2511F53u;
    unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < 1024; ++i) {
        REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\nv_mad_u64_u32 %1, vcc, %5, %8, 0\nv_mad_u64_u32 %2, vcc, %6, %8, 0\nv_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                          "v_mad_u64_u32 %0, vcc, %4, %8, 0\nv_mad_u64_u32 %1, vcc, %5, %8, 0\nv_mad_u64_u32 %2, vcc, %6, %8, 0\nv_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(d0 ^ d1 ^ d2 ^ d3);
}

template <typename K> static void run(const char* name, K kern, float* d_out) {
    printf("%-22s", name);
    for (int tb : {256, 512, 1024}) {            // 1 / 2 / 4 wavefronts per SIMD, one workgroup per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(tb), 0, 0, d_out);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(tb), 0, 0, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst = 1024.0 * 64.0 * (tb / 256);          // wave-instructions per SIMD
        printf("  %d w/SIMD: %6.2f cyc", tb / 256, ms * 1e-3 * 2.4e9 / inst);
    }
    printf("\n");
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 256 * 1024 * sizeof(float));
#define RUN(K) run(#K, K, d_out);
    RUN(k_add_f32) RUN(k_mul_f32) RUN(k_fma_f32) RUN(k_fma_f32_dep) RUN(k_max_f32) RUN(k_rndne_f32)
    RUN(k_log_f32) RUN(k_sin_f32) RUN(k_sqrt_f32) RUN(k_rcp_f32)
    RUN(k_pk_add_f32) RUN(k_pk_add_f32_swz) RUN(k_pk_mul_f32) RUN(k_pk_mul_f32_swz) RUN(k_pk_fma_f32) RUN(k_pk_fma_f32_swz)
    RUN(k_pk_fma_f32_dep) RUN(k_pk_mov_b32)
    RUN(k_mul_lo_u32) RUN(k_mul_hi_u32) RUN(k_mad_u64_u32) RUN(k_mul_u32_u24) RUN(k_mad_u32_u24)
    RUN(k_xor_b32) RUN(k_bitop3) RUN(k_add3_u32) RUN(k_lshl_add_u32) RUN(k_cndmask)
    RUN(k_cndmask_sgpr) RUN(k_cmp_cndmask) RUN(k_cmp_u32) RUN(k_cndmask_xor) RUN(k_cndmask_dst)
    RUN(k_cvt_f32_u32) RUN(k_cvt_i32_f32) RUN(k_fmac_f32) RUN(k_sub_f32) RUN(k_cos_f32)
    RUN(k_lshrrev) RUN(k_and_b32) RUN(k_add_u32) RUN(k_bfe_u32) RUN(k_alignbit) RUN(k_perm)
    return 0;
}
