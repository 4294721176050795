// Test-only probe of csrc/walk_f64.hpp::walk_decide<double, DEC, 4> -- the decision forms of the packed walk and of config 4's
// complex128 family -- on points the TEST chooses (tests/test_gpu_decide_probe.py compiles this file with hipcc at test time and
// loads it next to libmcle.so).  The pipelines draw their estimates, so the branch that serves a symbol the certificate does not
// vouch for (within 2^-30 of a decision boundary, or beyond the certificate's range) runs once in ~1e8 symbols there; here every
// probe point can be such a symbol.  Not part of the product: nothing under pyphysim_amd/ refers to it.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"
#include "modem.hpp"
#include "pipe_common.hpp"
#include "walk_f64.hpp"

using namespace mcle;

// one thread = one group of four estimates: its symbol / bit error counts against the four labels in tx
template <int DEC>
__global__ void k_probe(ModemParams<double> mp, const double2* __restrict__ pts, const int* __restrict__ tx, int n_groups,
                        unsigned* __restrict__ se_out, unsigned* __restrict__ be_out) {
    extern __shared__ __attribute__((aligned(16))) char probe_smem[];
    double2* s_table = reinterpret_cast<double2*>(probe_smem);
    unsigned long long* s_grid = reinterpret_cast<unsigned long long*>(s_table + ((mp.M + 1) & ~1));
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    __syncthreads();
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    double2 e[4];
    int t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = pts[4 * g + i];
        t[i] = tx[4 * g + i];
    }
    unsigned se = 0, be = 0;
    walk_decide<double, DEC, 4>(mp, s_table, s_grid, e, t, se, be);
    se_out[g] = se;
    be_out[g] = be;
}

// host: returns the decision form the library would compile for this context and method (WDEC_*), or -1000 - the HIP error
extern "C" int probe_walk_decide(mcle_ctx* ctx, int method, const double* pts_host, const int* tx_host, int n_groups,
                                 unsigned* se_host, unsigned* be_host) {
    ModemParams<double> mp = pipe_modem<double>(ctx, method);
    const int dec = walk_dec_kind(ctx, mp);
    double2* d_pts = nullptr;
    int* d_tx = nullptr;
    unsigned *d_se = nullptr, *d_be = nullptr;
    const size_t n = (size_t)4 * n_groups;
    hipError_t err;
#define PROBE_HIP(call) \
    if ((err = (call)) != hipSuccess) return -1000 - (int)err
    PROBE_HIP(hipMalloc(&d_pts, n * sizeof(double2)));
    PROBE_HIP(hipMalloc(&d_tx, n * sizeof(int)));
    PROBE_HIP(hipMalloc(&d_se, n_groups * sizeof(unsigned)));
    PROBE_HIP(hipMalloc(&d_be, n_groups * sizeof(unsigned)));
    PROBE_HIP(hipMemcpy(d_pts, pts_host, n * sizeof(double2), hipMemcpyHostToDevice));
    PROBE_HIP(hipMemcpy(d_tx, tx_host, n * sizeof(int), hipMemcpyHostToDevice));
    const size_t lds = (((size_t)mp.M + 1) & ~(size_t)1) * sizeof(double2) + ((size_t)mp.grid.G * mp.grid.G + 2) * sizeof(unsigned long long);
    const dim3 grid((unsigned)((n_groups + 255) / 256)), block(256);
    switch (dec) {
        case WDEC_GENERIC: hipLaunchKernelGGL(k_probe<WDEC_GENERIC>, grid, block, lds, 0, mp, d_pts, d_tx, n_groups, d_se, d_be); break;
        case WDEC_SLICER: hipLaunchKernelGGL(k_probe<WDEC_SLICER>, grid, block, lds, 0, mp, d_pts, d_tx, n_groups, d_se, d_be); break;
        case WDEC_QAM_CERT: hipLaunchKernelGGL(k_probe<WDEC_QAM_CERT>, grid, block, lds, 0, mp, d_pts, d_tx, n_groups, d_se, d_be); break;
        case WDEC_QUAD_CERT: hipLaunchKernelGGL(k_probe<WDEC_QUAD_CERT>, grid, block, lds, 0, mp, d_pts, d_tx, n_groups, d_se, d_be); break;
        default: hipLaunchKernelGGL(k_probe<WDEC_AXIS4_CERT>, grid, block, lds, 0, mp, d_pts, d_tx, n_groups, d_se, d_be); break;
    }
    PROBE_HIP(hipGetLastError());
    PROBE_HIP(hipDeviceSynchronize());
    PROBE_HIP(hipMemcpy(se_host, d_se, n_groups * sizeof(unsigned), hipMemcpyDeviceToHost));
    PROBE_HIP(hipMemcpy(be_host, d_be, n_groups * sizeof(unsigned), hipMemcpyDeviceToHost));
    (void)hipFree(d_pts);
    (void)hipFree(d_tx);
    (void)hipFree(d_se);
    (void)hipFree(d_be);
#undef PROBE_HIP
    return dec;
}
