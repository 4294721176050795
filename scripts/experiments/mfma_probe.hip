// mfma_probe.hip -- empirical lane / register maps of the two f32 MFMA forms the fused kernels use.
// build: hipcc -O2 --offload-arch=gfx950 scripts/experiments/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.
// Prints, for v_mfma_f32_16x16x4_f32 and v_mfma_f32_4x4x1_16b_f32, whether
//   16x16x4 : A[i][k] <- lane i + 16k,  B[k][j] <- lane j + 16k,  D[row][col] -> lane col + 16*(row/4), reg row%4
//   4x4x1   : A_b[i]  <- lane 4b + i,   B_b[j]  <- lane 4b + j,   D_b[i][j]   -> lane 4b + j, reg i
// hold (the maps csrc/fft16_mfma.hpp and the channel / decode stages are written against).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k16(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void k4(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }   // asymmetric
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    int bad = 0;
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col) {
            float want = 0;
            for (int k = 0; k < 4; ++k) want += ha[row + 16 * k] * hb[col + 16 * k];
            const float got = hd[(col + 16 * (row / 4)) * 4 + (row % 4)];
            if (got != want) ++bad;
        }
    printf("16x16x4 map mismatches: %d\n", bad);
    int bad4 = 0;
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const float want = ha[4 * blk + i] * hb[4 * blk + j];
                const float got = hd[(4 * blk + j) * 4 + i];
                if (got != want) ++bad4;
            }
    printf("4x4x1 map mismatches: %d\n", bad4);
    if (bad4) {   // dump enough to derive the real map
        for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[4*l], hd[4*l+1], hd[4*l+2], hd[4*l+3]);
    }
    return (bad || bad4) ? 1 : 0;
}
