// How much dynamic LDS may a 256-thread workgroup ask for and still share a CU with a second one?  Each workgroup
// spins for a fixed number of clock ticks; a grid of 2 x CUs workgroups takes one spin when two fit, two when not.
// build: hipcc -O3 --offload-arch=gfx950 lds_occupancy_probe.hip -o /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k_spin(long long ticks, int* sink) {
    extern __shared__ int s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (s[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* sink;
    hipMalloc(&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds = 76 * 1024; lds <= 82 * 1024; lds += 256) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_spin, dim3(2 * p.multiProcessorCount), dim3(256), lds, 0, 100000LL, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("lds %6d B: %.3f ms\n", lds, ms);
    }
    return 0;
}
