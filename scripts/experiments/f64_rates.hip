// f64_rates.hip -- what the complex128 pipelines can expect from gfx950's double-precision datapath:
//   (1) issue cost of the f64 VALU ops (ns per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD),
//   (2) issue cost and dependent latency of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64,
//   (3) whether an f64 MFMA stream runs beside a VALU stream (integer or f64) on the same SIMD,
//   (4) the operand / result lane maps of both f64 MFMA forms (hypothesis check + one-hot dump on mismatch).
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 scripts/experiments/f64_rates.hip -o /tmp/f64_rates && /tmp/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                            \
    __global__ void NAME(double* out) {                                                            \
        double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3, a4 = a0 + 4e-3,    \
               a5 = a0 + 5e-3, a6 = a0 + 6e-3, a7 = a0 + 7e-3, k = 1.0000001;                       \
        for (int i = 0; i < 1024; ++i) { REP8(ASM) }                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;        \
    }
#define A8(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
#define U8(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define F8(OP) asm volatile(OP " %0, %0, %8, %0\n" OP " %1, %1, %8, %1\n" OP " %2, %2, %8, %2\n" OP " %3, %3, %8, %3\n" OP " %4, %4, %8, %4\n" OP " %5, %5, %8, %5\n" OP " %6, %6, %8, %6\n" OP " %7, %7, %8, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
// dependent chain: one register only
#define DEP8(OP) asm volatile(OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" OP " %0, %0, %1, %0\n" \
    : "+v"(a0) : "v"(k));

BODY(k_add_f64, A8("v_add_f64"))
BODY(k_mul_f64, A8("v_mul_f64"))
BODY(k_fma_f64, F8("v_fma_f64"))
BODY(k_fma_f64_dep, DEP8("v_fma_f64"))
BODY(k_rcp_f64, U8("v_rcp_f64"))
BODY(k_rsq_f64, U8("v_rsq_f64"))
BODY(k_sqrt_f64, U8("v_sqrt_f64"))
BODY(k_floor_f64, U8("v_floor_f64"))
BODY(k_fract_f64, U8("v_fract_f64"))
BODY(k_max_f64, A8("v_max_f64"))
#define LDEXP8 asm volatile("v_ldexp_f64 %0, %0, %8\nv_ldexp_f64 %1, %1, %8\nv_ldexp_f64 %2, %2, %8\nv_ldexp_f64 %3, %3, %8\nv_ldexp_f64 %4, %4, %8\nv_ldexp_f64 %5, %5, %8\nv_ldexp_f64 %6, %6, %8\nv_ldexp_f64 %7, %7, %8\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0));
BODY(k_ldexp_f64, LDEXP8)
// conversions u32 -> f64 and f64 -> f32 round trip pieces
__global__ void k_cvt_f64_u32(double* out) {
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    for (int i = 0; i < 1024; ++i) {
        REP8(asm volatile("v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %9\nv_cvt_f64_u32 %2, %10\nv_cvt_f64_u32 %3, %11\nv_cvt_f64_u32 %4, %8\nv_cvt_f64_u32 %5, %9\nv_cvt_f64_u32 %6, %10\nv_cvt_f64_u32 %7, %11\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                          : "v"(u0), "v"(u1), "v"(u2), "v"(u3));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// 64-bit select = 2 v_cndmask_b32
__global__ void k_cmp_sel_f64(double* out) {
    double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3, k = 1.0000001;
    for (int i = 0; i < 1024; ++i) {
        REP8(asm volatile("v_cmp_lt_f64 vcc, %0, %4\nv_cmp_lt_f64 vcc, %1, %4\nv_cmp_lt_f64 vcc, %2, %4\nv_cmp_lt_f64 vcc, %3, %4\n"
                          "v_cmp_lt_f64 vcc, %0, %4\nv_cmp_lt_f64 vcc, %1, %4\nv_cmp_lt_f64 vcc, %2, %4\nv_cmp_lt_f64 vcc, %3, %4\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k) : "vcc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

// ---- MFMA issue / latency ----------------------------------------------------------------------------------------
template <int KIND, int CHAINS>   // KIND 0: 16x16x4 f64, 1: 4x4x4 f64 ; CHAINS independent accumulators
__global__ void k_mfma(double* out, int iters) {
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    d4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double e[4] = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (KIND == 0) c[q % CHAINS] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q % CHAINS], 0, 0, 0);
            else e[q % CHAINS] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e[q % CHAINS], 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + e[0] + e[1] + e[2] + e[3];
}

// ---- MFMA beside VALU on one SIMD: 512-thread blocks = 2 waves per SIMD ------------------------------------------
template <int MODE, int KIND, int VK>   // MODE bit 0: mfma waves run, bit 1: valu waves run; VK 0: int VALU, 1: f64 VALU
__global__ __launch_bounds__(512) void k_pair(double* out, int iters) {
    const int wave = threadIdx.x >> 6;
    double r = 0;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            } else {
                e0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e2, 0, 0, 0);
                e3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, e3, 0, 0, 0);
            }
        }
        r = c0[0] + c1[1] + c2[2] + c3[3] + e0 + e1 + e2 + e3;
    } else {
        if (!(MODE & 2)) return;
        if (VK == 0) {
            unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
            const unsigned k = 0x9E3779B9u;
            for (int i = 0; i < iters; ++i) {
                asm volatile("v_xor_b32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                             "v_xor_b32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                             "v_xor_b32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_xor_b32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                             "v_xor_b32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_xor_b32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            }
            r = (double)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
        } else {
            double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3, a4 = a0 + 4e-3,
                   a5 = a0 + 5e-3, a6 = a0 + 6e-3, a7 = a0 + 7e-3, k = 1.0000001;
            for (int i = 0; i < iters; ++i) {
                F8("v_fma_f64")
                F8("v_fma_f64")
            }
            r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// one wave per SIMD: K f64 FMAs (independent) after every 16x16x4 f64 MFMA
template <int K>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0;
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6, k = 1.0000001;
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[q & 7]) : "v"(k));
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[q & 7]) : "v"(k));
    }
    double r = c0[0] + c1[1];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- lane maps ---------------------------------------------------------------------------------------------------
__global__ void k_map16(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void k_map4(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

static float time_launch(void (*fn)(double*), int blocks, int threads, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <typename F> static float time_it(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* out;
    hipMalloc(&out, 8 << 20);
    struct K { const char* name; void (*fn)(double*); double per_iter; };
    K ks[] = {{"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64}, {"v_fma_f64", k_fma_f64, 64},
              {"v_fma_f64 (dependent)", k_fma_f64_dep, 64}, {"v_max_f64", k_max_f64, 64},
              {"v_rcp_f64", k_rcp_f64, 64}, {"v_rsq_f64", k_rsq_f64, 64}, {"v_sqrt_f64", k_sqrt_f64, 64},
              {"v_floor_f64", k_floor_f64, 64}, {"v_fract_f64", k_fract_f64, 64}, {"v_ldexp_f64", k_ldexp_f64, 64},
              {"v_cvt_f64_u32", k_cvt_f64_u32, 64}, {"v_cmp_lt_f64", k_cmp_sel_f64, 64}};
    printf("== (1) f64 VALU issue cost, every SIMD busy: ns per wave-instruction per SIMD (x 2.4 = cycles at 2.4 GHz)\n");
    for (auto& k : ks) {
        printf("%-24s", k.name);
        for (int wps : {1, 2, 4}) {
            const float ms = time_launch(k.fn, 256 * wps, 256, out);
            printf("  %d w/SIMD: %6.2f", wps, ms * 1e6 / (1024.0 * k.per_iter * wps));
        }
        printf("\n");
    }
    printf("== (2) f64 MFMA: ns per instruction per SIMD\n");
    const int iters = 4096;
    auto run_mfma = [&](const char* name, auto fn) {
        printf("%-40s", name);
        for (int wps : {1, 2, 4}) {
            const float ms = time_it([&]() { hipLaunchKernelGGL(fn, dim3(256 * wps), dim3(256), 0, 0, out, iters); });
            printf("  %d w/SIMD: %7.2f", wps, ms * 1e6 / (iters * 4.0 * wps));
        }
        printf("\n");
    };
    run_mfma("v_mfma_f64_16x16x4 (4 chains)", k_mfma<0, 4>);
    run_mfma("v_mfma_f64_16x16x4 (2 chains)", k_mfma<0, 2>);
    run_mfma("v_mfma_f64_16x16x4 (1 chain = latency)", k_mfma<0, 1>);
    run_mfma("v_mfma_f64_4x4x4_4b (4 chains)", k_mfma<1, 4>);
    run_mfma("v_mfma_f64_4x4x4_4b (1 chain = latency)", k_mfma<1, 1>);
    printf("== (3) MFMA waves beside VALU waves on the same SIMD (2 waves per SIMD): ms for mfma alone / valu alone / both\n");
    auto pair3 = [&](const char* name, auto f1, auto f2, auto f3) {
        const int it = 8192;
        const float m1 = time_it([&]() { hipLaunchKernelGGL(f1, dim3(256), dim3(512), 0, 0, out, it); });
        const float m2 = time_it([&]() { hipLaunchKernelGGL(f2, dim3(256), dim3(512), 0, 0, out, it); });
        const float m3 = time_it([&]() { hipLaunchKernelGGL(f3, dim3(256), dim3(512), 0, 0, out, it); });
        printf("%-44s mfma %.3f  valu %.3f  both %.3f  (sum %.3f, max %.3f)\n", name, m1, m2, m3, m1 + m2, m1 > m2 ? m1 : m2);
    };
    pair3("16x16x4 f64 + integer VALU", k_pair<1, 0, 0>, k_pair<2, 0, 0>, k_pair<3, 0, 0>);
    pair3("16x16x4 f64 + v_fma_f64", k_pair<1, 0, 1>, k_pair<2, 0, 1>, k_pair<3, 0, 1>);
    pair3("4x4x4 f64 + integer VALU", k_pair<1, 1, 0>, k_pair<2, 1, 0>, k_pair<3, 1, 0>);
    pair3("4x4x4 f64 + v_fma_f64", k_pair<1, 1, 1>, k_pair<2, 1, 1>, k_pair<3, 1, 1>);
    printf("== (3b) one wave per SIMD, K v_fma_f64 after every 16x16x4 f64 MFMA: ns per (MFMA + K FMAs)\n");
    {
        const int it = 8192;
        auto mix = [&](int k, auto fn) {
            const float ms = time_it([&]() { hipLaunchKernelGGL(fn, dim3(256), dim3(256), 0, 0, out, it); });
            printf("  K=%d: %.2f", k, ms * 1e6 / (it * 2.0));
        };
        mix(0, k_mix<0>); mix(2, k_mix<2>); mix(4, k_mix<4>); mix(8, k_mix<8>); mix(16, k_mix<16>);
        printf("\n");
    }
    // ---- (4) maps
    printf("== (4) lane maps\n");
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 2048);
    std::vector<double> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + l; hb[l] = 100.0 + 3 * l; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_map16, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col) {
            double want = 0;
            for (int k = 0; k < 4; ++k) want += ha[row + 16 * k] * hb[col + 16 * k];
            const double got = hd[(col + 16 * (row & 3)) * 4 + (row >> 2)];     // lane col + 16 (row % 4), reg row / 4
            if (got != want) ++bad;
        }
    printf("16x16x4 f64: A[i][k] <- lane i+16k, B[k][j] <- lane j+16k, D[row][col] -> lane col+16(row%%4), reg row/4: %d mismatches\n", bad);
    hipLaunchKernelGGL(k_map4, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
    // hypotheses for the 4-block form
    struct Hyp { const char* name; int (*al)(int, int, int); int (*bl)(int, int, int); int (*dl)(int, int, int); };
    Hyp hyps[] = {
        {"A_b[i][k] <- lane 16b+4k+i, B_b[k][j] <- lane 16b+4k+j, D_b[i][j] -> lane 16b+4i+j",
         [](int b, int i, int k) { return 16 * b + 4 * k + i; }, [](int b, int k, int j) { return 16 * b + 4 * k + j; },
         [](int b, int i, int j) { return 16 * b + 4 * i + j; }},
        {"A_b[i][k] <- lane 4b+i+16k, B_b[k][j] <- lane 4b+j+16k, D_b[i][j] -> lane 4b+j+16i",
         [](int b, int i, int k) { return 4 * b + i + 16 * k; }, [](int b, int k, int j) { return 4 * b + j + 16 * k; },
         [](int b, int i, int j) { return 4 * b + j + 16 * i; }},
        {"A_b[i][k] <- lane 4b+i+16k, B_b[k][j] <- lane 4b+j+16k, D_b[i][j] -> lane 16b+4i+j",
         [](int b, int i, int k) { return 4 * b + i + 16 * k; }, [](int b, int k, int j) { return 4 * b + j + 16 * k; },
         [](int b, int i, int j) { return 16 * b + 4 * i + j; }},
        {"A_b[i][k] <- lane 16b+4k+i, B_b[k][j] <- lane 16b+4k+j, D_b[i][j] -> lane 4b+j+16i",
         [](int b, int i, int k) { return 16 * b + 4 * k + i; }, [](int b, int k, int j) { return 16 * b + 4 * k + j; },
         [](int b, int i, int j) { return 4 * b + j + 16 * i; }},
    };
    for (auto& h : hyps) {
        int bad4 = 0;
        for (int blk = 0; blk < 4; ++blk)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    double want = 0;
                    for (int k = 0; k < 4; ++k) want += ha[h.al(blk, i, k)] * hb[h.bl(blk, k, j)];
                    if (hd[h.dl(blk, i, j)] != want) ++bad4;
                }
        printf("4x4x4 f64: %s: %d mismatches\n", h.name, bad4);
    }
    // one-hot dump: A = 1 at lane p only, B = 1 + lane  =>  D[l] = 1 + (B lane paired with A lane p at output l)
    printf("4x4x4 f64 one-hot map (A lane p: output lane<-B lane ...):\n");
    for (int p = 0; p < 64; ++p) {
        std::vector<double> oa(64, 0.0), ob(64);
        oa[p] = 1.0;
        for (int l = 0; l < 64; ++l) ob[l] = 1.0 + l;
        hipMemcpy(a, oa.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(b, ob.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_map4, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("  p=%2d:", p);
        for (int l = 0; l < 64; ++l)
            if (hd[l] != 0.0) printf(" %d<-%d", l, (int)hd[l] - 1);
        printf("\n");
    }
    return 0;
}
