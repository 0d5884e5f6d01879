// Microbenchmark: issue rate of the VALU instructions used by the Chamfer hot loop (gfx950).
// Each kernel runs ITER iterations of 32 independent instructions per wave; waves/SIMD is a launch parameter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define DEF_KERNEL(name, body)                                                  \
    __global__ __launch_bounds__(256) void name(float* out, int n) {            \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        typedef float f2 __attribute__((ext_vector_type(2)));                   \
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}; \
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;             \
        for (int it = 0; it < n; ++it) { body }                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + i0 + i1 + i2 + i3; \
    }
// 32 instructions per iteration
DEF_KERNEL(k_add, REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
DEF_KERNEL(k_fma, REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));))
DEF_KERNEL(k_pk_add, REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4));))
DEF_KERNEL(k_pk_mul, REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4));))
DEF_KERNEL(k_pk_fma, REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4), "v"(p5));))
DEF_KERNEL(k_min, REP8(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
DEF_KERNEL(k_min3, REP8(asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));))
DEF_KERNEL(k_cmp, REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");))
DEF_KERNEL(k_cmp_sgpr, REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");))
DEF_KERNEL(k_cnd, REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : );))
// the update pattern of the kernel: cmp + 3 cndmask on one chain, 8 per iteration (4 chains x 2)
DEF_KERNEL(k_update, REP8(asm volatile("v_cmp_lt_f32 vcc, %4, %0\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %2, %2, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(i0) : "v"(0), "v"(a4), "v"(i1) : "vcc");))
DEF_KERNEL(k_update2, REP8(asm volatile("v_cmp_lt_f32 vcc, %6, %0\n v_cmp_lt_f32 s[20:21], %6, %3\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %2, %2, %7, vcc\n v_cndmask_b32 %4, %4, %3, s[20:21]\n v_cndmask_b32 %3, %3, %6, s[20:21]\n v_cndmask_b32 %5, %5, %7, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(i0), "+v"(a2), "+v"(a3), "+v"(i2) : "v"(a4), "v"(i1) : "vcc", "s20", "s21");))
DEF_KERNEL(k_med3, REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));))
DEF_KERNEL(k_mix, REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_add_f32 %2, %2, %3\n v_pk_mul_f32 %1, %1, %4\n v_add_f32 %3, %3, %2" : "+v"(p0), "+v"(p1), "+v"(a0), "+v"(a1) : "v"(p4));))

template <typename K>
void run(const char* name, K kern, int instr_per_iter, float* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {                     // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        int blocks = 256 * wps;                        // 256 CUs x wps blocks
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, ITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)ITER * instr_per_iter * wps;      // wave-instructions issued on one SIMD
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD = %.2f cyc @2.4GHz\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run("v_add", k_add, 32, d); run("v_fma", k_fma, 32, d); run("pk_add", k_pk_add, 32, d); run("pk_mul", k_pk_mul, 32, d);
    run("pk_fma", k_pk_fma, 32, d); run("v_min", k_min, 32, d); run("v_min3", k_min3, 32, d); run("v_med3", k_med3, 32, d);
    run("cmp_vcc", k_cmp, 32, d); run("cmp_sgpr", k_cmp_sgpr, 32, d); run("cndmask", k_cnd, 32, d);
    run("update", k_update, 32, d); run("update2", k_update2, 64, d); run("mix", k_mix, 32, d);
    return 0;
}
