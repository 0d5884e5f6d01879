// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal INPUTS (gfx950)?  A[i][k] = 2^-20 (subnormal in fp16), B[k][j] = 2^10
// for k = 0 only: exact product 2^-10 per output if subnormals are honoured, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    float h[64];
    const float tests[][2] = {{9.5367431640625e-07f, 1024.f}, {1.0f, 1.0f}, {6.103515625e-05f, 2.0f}, {5.9604644775390625e-08f, 16384.f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> out[0]=%g (exact %g)\n", t[0], t[1], h[0], (double)t[0] * t[1]);
    }
    return 0;
}
