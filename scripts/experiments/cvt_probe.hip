// rounding / saturation of v_cvt_pk_u8_f32 on gfx950 (used by the packed QAM slicer)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
    float h[16] = {0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 2.7f, -3.f, 300.f, 6.99f, 7.0f, 7.49f, 7.5f, 254.6f, -0.4f, 1e9f};
    float* d; unsigned* o; unsigned ho[16];
    hipMalloc(&d, 64); hipMalloc(&o, 64);
    hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 16);
    hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%g -> %u\n", h[i], ho[i]);
    return 0;
}
