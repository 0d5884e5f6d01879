// Microbenchmark: issue rate of the fp32-input MFMA instructions on gfx950 (waves/SIMD as a parameter).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define ITER 2048
__global__ __launch_bounds__(256) void k32(float* out, int n) {
    f16v a = {0}, b = {0}, c = {0}, d = {0};
    float x = threadIdx.x, y = x + 1;
    for (int i = 0; i < n; ++i) {
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, d, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + b[1] + c[2] + d[3];
}
__global__ __launch_bounds__(256) void k32zero(float* out, int n) {   // C = 0 each time + dependent second (our pattern)
    f16v z = {0};
    float x = threadIdx.x, y = x + 1, s = 0;
    for (int i = 0; i < n; ++i) {
        f16v a = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, z, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a, 0, 0, 0);
        f16v b = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, z, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, b, 0, 0, 0);
        s += a[0] + b[5];
        x += 1e-9f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(float* out, int n) {
    f4v a = {0}, b = {0}, c = {0}, d = {0};
    float x = threadIdx.x, y = x + 1;
    for (int i = 0; i < n; ++i) {
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, d, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a[0] + b[1] + c[2] + d[3];
}
template <typename K> void run(const char* name, K kern, double flop_per_instr, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)ITER * 4 * wps;
        printf("%-8s waves/SIMD=%d %.3f ms -> %.1f ns per MFMA per SIMD = %.1f cyc@2.4GHz ; %.1f TFLOP/s\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, flop_per_instr * instr_per_simd * 1024 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run("32x32x2", k32, 4096.0, d);
    run("32x32x2z", k32zero, 4096.0, d);
    run("16x16x4", k16, 2048.0, d);
    return 0;
}
