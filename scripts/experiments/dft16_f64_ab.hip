// dft16_f64_ab.hip -- the building block of a complex128 FFT-1024 = 16 x 16 x 4, two ways, on one MI355X:
//   (A) VALU : a DFT-16 of every group as two radix-4 butterfly stages (v_add_f64 / v_mul_f64 / v_fma_f64), the form
//              k_run_mimo_ofdm_f64 uses;
//   (B) MFMA : the same DFT-16 as the real matrix product of csrc/fft16.hpp ported to v_mfma_f64_16x16x4_f64
//              (radix-2 split on the VALU, two 16x16x16 real products = 8 MFMAs per 16 groups, D map of the f64 form:
//              row = (lane >> 4) + 4 reg).
// Both transform 4 antennas x 64 groups x 16 points per workgroup pass (one "DFT-16 pass" of the fused kernel, followed by
// the pass's twiddle multiplication), data planar in LDS as [antenna][re | im][element e][group] so that every access of both
// variants is a contiguous, bank-conflict-free run.  Results are checked against a host DFT; the timing loop repeats the
// pass in place.  Two workgroups of 256 threads per CU (the fused kernel's occupancy).
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/experiments/dft16_f64_ab.hip -o /tmp/dft16_ab && /tmp/dft16_ab
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NA = 4, NG = 64, NE = 16;                 // antennas, groups per antenna, points per group
constexpr int PLANE = NE * NG;                          // doubles per plane
__device__ __forceinline__ int at(int a, int part, int e, int grp) { return ((a * 2 + part) * NE + e) * NG + grp; }

// twiddle of output k of group grp (any per-element factor does: the fused kernel multiplies by W1024^{k n2} here)
__host__ __device__ inline void post_tw(int k, int grp, double& c, double& s) {
    const double ang = -2.0 * 3.14159265358979323846 * (double)(k * grp) / 1024.0;
    c = cos(ang);
    s = sin(ang);
}

// ---- (A) two radix-4 stages per group on the VALU; thread = (butterfly i = tid / 64, group = tid % 64) ----------------
__global__ __launch_bounds__(256, 2) void k_valu(const double* __restrict__ in, double* __restrict__ out, int iters) {
    extern __shared__ double s[];
    const int tid = threadIdx.x, i = tid >> 6, grp = tid & 63;
    for (int p = tid; p < NA * 2 * PLANE; p += 256) s[p] = in[(size_t)blockIdx.x * NA * 2 * PLANE + p];
    // stage-1 twiddles W16^{i q}, q = 1..3 and the post twiddles of this thread's four outputs of stage 2
    double w1c[3], w1s[3], pc[4], psn[4];
    for (int q = 1; q < 4; ++q) {
        const double ang = -2.0 * 3.14159265358979323846 * (double)(i * q) / 16.0;
        w1c[q - 1] = cos(ang);
        w1s[q - 1] = sin(ang);
    }
    for (int q = 0; q < 4; ++q) post_tw(4 * q + i, grp, pc[q], psn[q]);   // stage 2 butterfly i produces outputs k = i + 4 q
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // stage 1 (DIF, span 4): elements i, i + 4, i + 8, i + 12; output q times W16^{i q}
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            double xr[4], xi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xr[q] = s[at(a, 0, i + 4 * q, grp)];
                xi[q] = s[at(a, 1, i + 4 * q, grp)];
            }
            const double a0r = xr[0] + xr[2], a0i = xi[0] + xi[2], a1r = xr[0] - xr[2], a1i = xi[0] - xi[2];
            const double a2r = xr[1] + xr[3], a2i = xi[1] + xi[3], a3r = xi[1] - xi[3], a3i = xr[3] - xr[1];   // (x1 - x3)(-j)
            const double yr[4] = {a0r + a2r, a1r + a3r, a0r - a2r, a1r - a3r};
            const double yi[4] = {a0i + a2i, a1i + a3i, a0i - a2i, a1i - a3i};
            s[at(a, 0, i, grp)] = yr[0];
            s[at(a, 1, i, grp)] = yi[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                s[at(a, 0, i + 4 * q, grp)] = yr[q] * w1c[q - 1] - yi[q] * w1s[q - 1];
                s[at(a, 1, i + 4 * q, grp)] = yr[q] * w1s[q - 1] + yi[q] * w1c[q - 1];
            }
        }
        __syncthreads();
        // stage 2 (span 1): elements 4 i .. 4 i + 3 hold the sub-sequence with first output digit i; output q -> X[i + 4 q]
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            double xr[4], xi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xr[q] = s[at(a, 0, 4 * i + q, grp)];
                xi[q] = s[at(a, 1, 4 * i + q, grp)];
            }
            const double a0r = xr[0] + xr[2], a0i = xi[0] + xi[2], a1r = xr[0] - xr[2], a1i = xi[0] - xi[2];
            const double a2r = xr[1] + xr[3], a2i = xi[1] + xi[3], a3r = xi[1] - xi[3], a3i = xr[3] - xr[1];
            const double yr[4] = {a0r + a2r, a1r + a3r, a0r - a2r, a1r - a3r};
            const double yi[4] = {a0i + a2i, a1i + a3i, a0i - a2i, a1i - a3i};
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // X[i + 4 q] x post twiddle, parked at element 4 i + q (digit-reversed)
                s[at(a, 0, 4 * i + q, grp)] = yr[q] * pc[q] - yi[q] * psn[q];
                s[at(a, 1, 4 * i + q, grp)] = yr[q] * psn[q] + yi[q] * pc[q];
            }
        }
        __syncthreads();
    }
    for (int p = tid; p < NA * 2 * PLANE; p += 256) out[(size_t)blockIdx.x * NA * 2 * PLANE + p] = s[p];
}

// ---- (B) the DFT-16 as two real 16 x 16 x 16 products on v_mfma_f64_16x16x4_f64; wave w owns groups 16 w .. 16 w + 15 ----
__global__ __launch_bounds__(256, 2) void k_mfma(const double* __restrict__ in, double* __restrict__ out, int iters) {
    extern __shared__ double s[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    for (int p = tid; p < NA * 2 * PLANE; p += 256) s[p] = in[(size_t)blockIdx.x * NA * 2 * PLANE + p];
    // A operands: lane (row i = lane & 15, k-group g): k-step t covers element e = 2 t + (g >> 1), part g & 1;
    // f64 D map: register x of lane g is row g + 4 x, so row i is (u = 2 (i & 3) + (i >> 3), part_out = (i >> 2) & 1)
    double ae[4], ao[4];
    {
        const int i = lane & 15, u = 2 * (i & 3) + (i >> 3), part_out = (i >> 2) & 1;
        for (int t = 0; t < 4; ++t) {
            const int e = 2 * t + (g >> 1);
            const double ange = -2.0 * 3.14159265358979323846 * (double)(e * u) / 8.0;
            const double ango = -2.0 * 3.14159265358979323846 * (double)(e * (2 * u + 1)) / 16.0;
            const double wer = cos(ange), wei = sin(ange), wor = cos(ango), woi = sin(ango);
            if (part_out == 0) {
                ae[t] = (g & 1) ? -wei : wer;
                ao[t] = (g & 1) ? -woi : wor;
            } else {
                ae[t] = (g & 1) ? wer : wei;
                ao[t] = (g & 1) ? wor : woi;
            }
        }
    }
    const int grp = 16 * w + j;
    double pc[4], psn[4];
    for (int x = 0; x < 4; ++x) post_tw(4 * g + x, grp, pc[x], psn[x]);     // this lane's outputs: k = 4 g + x of group grp
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        double b[NA][8];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int t = 0; t < 8; ++t) b[a][t] = s[at(a, g & 1, 2 * (t & 3) + (g >> 1) + 8 * (t >> 2), grp)];
        d4 ce[NA], co[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) ce[a] = co[a] = d4{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                ce[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[t], b[a][t] + b[a][t + 4], ce[a], 0, 0, 0);
                co[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[t], b[a][t] - b[a][t + 4], co[a], 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();            // in place inside the wavefront: every lane holds its operands
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const double orr[4] = {ce[a][0], co[a][0], ce[a][2], co[a][2]};
            const double oi[4] = {ce[a][1], co[a][1], ce[a][3], co[a][3]};
#pragma unroll
            for (int x = 0; x < 4; ++x) {           // X[4 g + x] x post twiddle -> element 4 g + x (natural order)
                s[at(a, 0, 4 * g + x, grp)] = orr[x] * pc[x] - oi[x] * psn[x];
                s[at(a, 1, 4 * g + x, grp)] = orr[x] * psn[x] + oi[x] * pc[x];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int p = tid; p < NA * 2 * PLANE; p += 256) out[(size_t)blockIdx.x * NA * 2 * PLANE + p] = s[p];
}

int main() {
    const int blocks = 512, per = NA * 2 * PLANE;
    std::vector<double> h((size_t)blocks * per);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        v = (double)(st >> 11) / 9007199254740992.0 - 0.5;
    }
    double *din, *dout;
    hipMalloc(&din, h.size() * 8);
    hipMalloc(&dout, h.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)per * 8;
    hipFuncSetAttribute((const void*)k_valu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<double> o((size_t)blocks * per);
    // ---- numerics: one pass of block 3 against the host DFT ----
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), lds, 0, din, dout, 1);
        else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), lds, 0, din, dout, 1);
        hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        const size_t base = (size_t)3 * per;
        for (int a = 0; a < NA; ++a)
            for (int grp = 0; grp < NG; ++grp)
                for (int k = 0; k < NE; ++k) {
                    std::complex<double> acc = 0;
                    for (int e = 0; e < NE; ++e) {
                        const std::complex<double> x(h[base + ((a * 2 + 0) * NE + e) * NG + grp], h[base + ((a * 2 + 1) * NE + e) * NG + grp]);
                        acc += x * std::polar(1.0, -2.0 * 3.14159265358979323846 * (double)(e * k) / 16.0);
                    }
                    double c, sn;
                    post_tw(k, grp, c, sn);
                    acc *= std::complex<double>(c, sn);
                    // where output k sits: (A) digit-reversed: k = i + 4 q at element 4 i + q; (B) natural
                    const int el = variant == 0 ? 4 * (k & 3) + (k >> 2) : k;
                    const std::complex<double> got(o[base + ((a * 2 + 0) * NE + el) * NG + grp], o[base + ((a * 2 + 1) * NE + el) * NG + grp]);
                    worst = fmax(worst, std::abs(got - acc));
                }
        printf("%s: max |error| of one pass against the host DFT-16: %.3e\n", variant == 0 ? "VALU radix-4 x 2" : "MFMA f64 16x16x4", worst);
    }
    // ---- timing ----
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            if (variant == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), lds, 0, din, dout, iters);
            else hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), lds, 0, din, dout, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        // all 512 workgroups are resident (two per CU), so a workgroup's pass takes best / iters
        printf("%s: %.3f ms for %d in-place passes of 512 resident workgroups: %.3f us per DFT-16 pass (4 antennas x 1024 points), "
               "%.2f us for the 16 passes of one realization's eight transforms\n",
               variant == 0 ? "VALU radix-4 x 2" : "MFMA f64 16x16x4", best, iters, best * 1e3 / iters, best * 1e3 / iters * 16.0);
    }
    return 0;
}
