// Does VALU work overlap with MFMA work on one SIMD?  fp32-input MFMA vs bf16 MFMA, each followed by the
// min-tree + bookkeeping the Chamfer filter does per 32x32 block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define ITER 4096
__device__ inline float min16(const f16v& v) {
    float a = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]), b = __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]);
    float c = __builtin_fminf(__builtin_fminf(v[6], v[7]), v[8]), d = __builtin_fminf(__builtin_fminf(v[9], v[10]), v[11]);
    float e = __builtin_fminf(__builtin_fminf(v[12], v[13]), v[14]);
    return __builtin_fminf(__builtin_fminf(__builtin_fminf(a, b), c), __builtin_fminf(__builtin_fminf(d, e), v[15]));
}
template <int MODE>   // 0: fp32 mfma + valu, 1: bf16 mfma + valu, 2: fp32 mfma only, 3: bf16 mfma only, 4: valu only
__global__ __launch_bounds__(256) void k(float* out, int n) {
    const f16v z = {0};
    float x = threadIdx.x * 1e-3f, y = x + 1, best = 1e30f, second = 1e30f; int blk = 0;
    bf8 p = {1, 2, 3, 4, 5, 6, 7, 8}, q = {8, 7, 6, 5, 4, 3, 2, 1};
    f16v acc = z, acc2 = z;
    for (int i = 0; i < n; ++i) {
        if (MODE == 0 || MODE == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, z, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc, 0, 0, 0);
        } else if (MODE == 1 || MODE == 3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, z, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, p, acc, 0, 0, 0);
        } else { acc[0] = x * y; acc[7] = x + y; asm volatile("" : "+v"(acc)); }
        if (MODE == 0 || MODE == 1 || MODE == 4) {
            const float m = min16(acc);
            const bool up = m < best;
            second = __builtin_amdgcn_fmed3f(best, second, m);
            blk = up ? i : blk; best = up ? m : best;
        } else { asm volatile("" :: "v"(acc)); best += acc[3]; }
        x += 1e-7f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = best + second + blk + acc2[0];
}
template <int MODE> void run(const char* name, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-18s waves/SIMD=%d  %.1f ns per block-iteration per SIMD\n", name, wps, ms * 1e6 / ((double)ITER * wps));
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<2>("fp32 mfma only", d); run<3>("bf16 mfma only", d); run<4>("valu only", d);
    run<0>("fp32 mfma + valu", d); run<1>("bf16 mfma + valu", d);
    return 0;
}
