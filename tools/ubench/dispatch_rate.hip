// How fast does the chip start workgroups?  The same number of waves as `wgs` workgroups of 1, 2, 4 or 8 waves; each wave
// either exits at once or walks a chain of `chain` dependent global loads (a tile wave's latency chain).  Time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void body(const int* __restrict__ p, int chain, int* sink) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) & 1023;
    for (int c = 0; c < chain; ++c) i = p[i];
    if (i == 0x7fffffff) *sink = i;
}
// the same with the tile kernel's footprint: 96 VGPRs (5 waves per SIMD) and 4.6 KB of dynamic LDS per one-wave workgroup
__global__ void body_fat(const int* __restrict__ p, int chain, int* sink) {
    extern __shared__ int lds[];
    int i = (blockIdx.x * blockDim.x + threadIdx.x) & 1023;
    asm volatile("v_mov_b32 v95, 0" ::: "v95");
    for (int c = 0; c < chain; ++c) i = p[i];
    if (i == 0x7fffffff) { lds[threadIdx.x] = i; *sink = lds[(threadIdx.x + 1) & 63]; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    int* p; int* sink;
    CK(hipMalloc(&p, 1024 * sizeof(int))); CK(hipMalloc(&sink, 4));
    int h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (i * 37 + 11) & 1023;
    CK(hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int waves : {16384, 65536}) {
        for (int chain : {0, 4}) {
            for (int wpw : {1, 2, 4, 8, 16}) {
                const int wgs = waves / wpw;
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(body, dim3(wgs), dim3(64 * wpw), 0, 0, p, chain, sink);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(body, dim3(wgs), dim3(64 * wpw), 0, 0, p, chain, sink);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%6d waves as %6d workgroups of %2d waves, chain of %d loads: %7.2f us per launch\n", waves, wgs, wpw, chain, ms * 1e3f / 20);
            }
        }
    }
    for (int waves : {16384, 65536}) {
        for (int chain : {0, 4, 8}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(body_fat, dim3(waves), dim3(64), 4640, 0, p, chain, sink);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(body_fat, dim3(waves), dim3(64), 4640, 0, p, chain, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%6d one-wave workgroups with 96 VGPRs + 4.6 KB LDS, chain of %d loads: %7.2f us per launch\n", waves, chain, ms * 1e3f / 20);
        }
    }
    return 0;
}
