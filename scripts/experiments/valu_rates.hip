// valu_rates.hip -- issue cost (cycles per wave64 instruction, one wave on its SIMD) of the VALU ops that dominate the
// fused kernels: what a Philox round, a Box-Muller sample and a butterfly really cost on gfx950.
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 scripts/experiments/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                            \
    __global__ void NAME(unsigned* out, long long* cyc) {                                         \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, k = 0x9E3779B9u;                                                     \
        unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                     \
        (void)w0; (void)w1; (void)w2; (void)w3;                                                    \
        long long t0 = __builtin_readcyclecounter();                                               \
        for (int i = 0; i < 2048; ++i) { REP8(ASM) }                                               \
        long long t1 = __builtin_readcyclecounter();                                               \
        out[threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(w0 ^ w1 ^ w2 ^ w3);  \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                    \
    }
// each ASM block = 8 independent instructions
#define A8(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#define U8(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define F8(OP) asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n" OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#define M8 asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\nv_mad_u64_u32 %1, vcc, %4, %6, 0\nv_mad_u64_u32 %2, vcc, %4, %7, 0\nv_mad_u64_u32 %3, vcc, %4, %8, 0\n" \
                        "v_mad_u64_u32 %0, vcc, %4, %5, 0\nv_mad_u64_u32 %1, vcc, %4, %6, 0\nv_mad_u64_u32 %2, vcc, %4, %7, 0\nv_mad_u64_u32 %3, vcc, %4, %8, 0\n" \
    : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(k), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
#define P8 asm volatile("v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n" \
                        "v_pk_fma_f32 %0, %0, %4, %0\nv_pk_fma_f32 %1, %1, %4, %1\nv_pk_fma_f32 %2, %2, %4, %2\nv_pk_fma_f32 %3, %3, %4, %3\n" \
    : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w0));
#define C8 asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
#define D8 asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
                        "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

BODY(k_xor, A8("v_xor_b32"))
BODY(k_add_u32, A8("v_add_u32"))
BODY(k_and, A8("v_and_b32"))
BODY(k_lshl, A8("v_lshlrev_b32"))
BODY(k_add_f32, A8("v_add_f32"))
BODY(k_mul_f32, A8("v_mul_f32"))
BODY(k_fma_f32, F8("v_fma_f32"))
BODY(k_pk_fma, P8)
BODY(k_mad_u64, M8)
BODY(k_mul_hi, A8("v_mul_hi_u32"))
BODY(k_mul_lo, A8("v_mul_lo_u32"))
BODY(k_log, U8("v_log_f32"))
BODY(k_sqrt, U8("v_sqrt_f32"))
BODY(k_sin, U8("v_sin_f32"))
BODY(k_cvt_f32_u32, U8("v_cvt_f32_u32"))
BODY(k_floor, U8("v_floor_f32"))
BODY(k_cndmask, C8)
BODY(k_dpp, D8)
BODY(k_bfe, F8("v_bfe_u32"))
#define B8 asm volatile("v_bitop3_b32 %0, %0, %8, %1 bitop3:0x96\nv_bitop3_b32 %1, %1, %8, %2 bitop3:0x96\nv_bitop3_b32 %2, %2, %8, %3 bitop3:0x96\nv_bitop3_b32 %3, %3, %8, %4 bitop3:0x96\nv_bitop3_b32 %4, %4, %8, %5 bitop3:0x96\nv_bitop3_b32 %5, %5, %8, %6 bitop3:0x96\nv_bitop3_b32 %6, %6, %8, %7 bitop3:0x96\nv_bitop3_b32 %7, %7, %8, %0 bitop3:0x96\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
BODY(k_xor3, B8)
BODY(k_perm, F8("v_perm_b32"))

int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    struct K { const char* name; void (*fn)(unsigned*, long long*); };
    K ks[] = {{"v_xor_b32", k_xor}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl},
              {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma},
              {"v_mad_u64_u32", k_mad_u64}, {"v_mul_hi_u32", k_mul_hi}, {"v_mul_lo_u32", k_mul_lo}, {"v_log_f32", k_log},
              {"v_sqrt_f32", k_sqrt}, {"v_sin_f32", k_sin}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_floor_f32", k_floor},
              {"v_cndmask_b32", k_cndmask}, {"v_mov_b32_dpp", k_dpp}, {"v_bfe_u32", k_bfe}, {"v_bitop3_b32 (xor3)", k_xor3},
              {"v_perm_b32", k_perm}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // every SIMD of the chip busy (clocks up): 256 CUs x W waves per SIMD; time = wall clock of the launch
    for (auto& k : ks) {
        printf("%-20s", k.name);
        for (int wps : {1, 2, 4}) {
            const int blocks = 256 * wps;          // 256-thread blocks = one wave per SIMD each
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, cyc);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, cyc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_simd = 2048.0 * 8 * 8 * wps;       // per wave x waves on the SIMD
            printf("  %d w/SIMD: %6.2f ns/inst/SIMD", wps, ms * 1e6 / insts_per_simd);
        }
        printf("\n");
    }
    return 0;
}
