// Microbenchmark 2: issue rate of integer min / compare, select, transcendental and DPP forms (gfx950), same
// harness as valu_rates.hip: ITER iterations of 32 independent instructions per wave, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define DEF_KERNEL(name, body)                                                  \
    __global__ __launch_bounds__(256) void name(float* out, int n) {            \
        float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5; \
        int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5; \
        for (int it = 0; it < n; ++it) { body }                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + i0 + i1 + i2 + i3 + i4 + i5; \
    }
#define Q4(op, x) asm volatile(op " %0, %0, %4" x "\n " op " %1, %1, %4" x "\n " op " %2, %2, %4" x "\n " op " %3, %3, %4" x
DEF_KERNEL(k_min_u32, REP8(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4));))
DEF_KERNEL(k_min_i32, REP8(asm volatile("v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_min_i32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4));))
DEF_KERNEL(k_min3_u32, REP8(asm volatile("v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_min3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4), "v"(i5));))
DEF_KERNEL(k_min3_i32, REP8(asm volatile("v_min3_i32 %0, %0, %4, %5\n v_min3_i32 %1, %1, %4, %5\n v_min3_i32 %2, %2, %4, %5\n v_min3_i32 %3, %3, %4, %5" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4), "v"(i5));))
DEF_KERNEL(k_med3_u32, REP8(asm volatile("v_med3_u32 %0, %0, %4, %5\n v_med3_u32 %1, %1, %4, %5\n v_med3_u32 %2, %2, %4, %5\n v_med3_u32 %3, %3, %4, %5" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4), "v"(i5));))
DEF_KERNEL(k_cmp_u32, REP8(asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %2\n v_cmp_lt_u32 s[24:25], %2, %3\n v_cmp_lt_u32 s[26:27], %3, %0" : : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");))
DEF_KERNEL(k_max_f32, REP8(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
DEF_KERNEL(k_mul_f32, REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
DEF_KERNEL(k_and_b32, REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4));))
DEF_KERNEL(k_add_u32, REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i4));))
DEF_KERNEL(k_cnd_sgpr, REP8(asm volatile("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "s20", "s21");))
DEF_KERNEL(k_exp, REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));))
DEF_KERNEL(k_rcp, REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));))
DEF_KERNEL(k_sqrt, REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));))
DEF_KERNEL(k_rsq, REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));))
DEF_KERNEL(k_fma_exp, REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));))
DEF_KERNEL(k_add_dpp, REP8(asm volatile("v_add_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));))
DEF_KERNEL(k_fma_sgpr, REP8(asm volatile("v_fma_f32 %0, %0, s20, %5\n v_fma_f32 %1, %1, s21, %5\n v_fma_f32 %2, %2, s22, %5\n v_fma_f32 %3, %3, s23, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5) : "s20", "s21", "s22", "s23");))
DEF_KERNEL(k_mac, REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));))

template <typename K>
void run(const char* name, K kern, int instr_per_iter, float* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, ITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)ITER * instr_per_iter * wps;
        printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD = %.2f cyc @2.4GHz\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run("min_u32", k_min_u32, 32, d); run("min_i32", k_min_i32, 32, d); run("min3_u32", k_min3_u32, 32, d); run("min3_i32", k_min3_i32, 32, d);
    run("med3_u32", k_med3_u32, 32, d); run("cmp_u32", k_cmp_u32, 32, d); run("max_f32", k_max_f32, 32, d); run("mul_f32", k_mul_f32, 32, d);
    run("and_b32", k_and_b32, 32, d); run("add_u32", k_add_u32, 32, d); run("cnd_sgpr", k_cnd_sgpr, 32, d);
    run("exp", k_exp, 32, d); run("rcp", k_rcp, 32, d); run("sqrt", k_sqrt, 32, d); run("rsq", k_rsq, 32, d); run("3fma+exp", k_fma_exp, 32, d);
    run("add_dpp", k_add_dpp, 32, d); run("fma_sgpr", k_fma_sgpr, 32, d); run("fmac", k_mac, 32, d);
    return 0;
}
