// Does an MFMA in flight on a SIMD overlap with VALU work (a) of ANOTHER wave on the same SIMD, (b) of the SAME wave
// when the VALU work is independent of it (software pipelining), (c) of the same wave when it depends on it (the
// Chamfer filter loop as written)?  512-thread workgroups: wave w sits on SIMD w & 3, so waves 0-3 and 4-7 pair up
// on the four SIMDs of a CU.  One workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define ITER 8192

__device__ inline float tree(const f16v& v) {          // 8 v_min3 as in chamfer.hip
    float a, b, c, d, e, f, g, h;
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(v[3]), "v"(v[4]), "v"(v[5]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(c) : "v"(v[6]), "v"(v[7]), "v"(v[8]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(v[9]), "v"(v[10]), "v"(v[11]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(e) : "v"(v[12]), "v"(v[13]), "v"(v[14]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(f) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(g) : "v"(d), "v"(e), "v"(v[15]));
    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(h) : "v"(f), "v"(g), "v"(g));
    return h;
}
#define BOOK(m)                                                                                   \
    {                                                                                             \
        unsigned long long k_;                                                                    \
        asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(k_) : "v"(m), "v"(best));                   \
        second = __builtin_amdgcn_fmed3f(best, second, m);                                        \
        asm volatile("v_cndmask_b32 %0, %1, 3, %2" : "=v"(blk) : "v"(blk), "s"(k_));              \
        asm volatile("v_min_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(m));                    \
    }

// role: 0 idle, 1 MFMA only, 2 VALU only (tree + bookkeeping on a register-resident vector), 3 dependent MFMA -> tree
// (the kernel's loop), 4 software pipelined (MFMA of block i+1 issued before the tree of block i)
template <int RLO, int RHI>
__global__ __launch_bounds__(512) void k(float* out, int n) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? RLO : RHI;
    const f16v z = {0};
    h8 p, q;
    for (int e = 0; e < 8; ++e) { p[e] = (_Float16)(threadIdx.x * 0.001f + e); q[e] = (_Float16)(e + 1); }
    float best = 1e30f, second = 1e30f; int blk = 0;
    f16v acc = z, acc2 = z;
    for (int e = 0; e < 16; ++e) acc[e] = threadIdx.x + e;
    if (role == 1) {
        for (int i = 0; i < n; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p, q, z, 0, 0, 0);
            asm volatile("" : "+v"(acc));
        }
        best = acc[3];
    } else if (role == 2) {
        for (int i = 0; i < n; ++i) {
            asm volatile("" : "+v"(acc));
            const float m = tree(acc);
            BOOK(m)
        }
    } else if (role == 3) {
        for (int i = 0; i < n; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p, q, z, 0, 0, 0);
            const float m = tree(acc);
            BOOK(m)
        }
    } else if (role == 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p, q, z, 0, 0, 0);
        for (int i = 0; i < n; i += 2) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, p, z, 0, 0, 0);
            { const float m = tree(acc); BOOK(m) }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p, q, z, 0, 0, 0);
            { const float m = tree(acc2); BOOK(m) }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = best + second + blk;
}
template <int RLO, int RHI> void run(const char* name, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wg : {1, 2, 3}) {                                 // workgroups per CU: 2, 4, 6 waves per SIMD
        hipLaunchKernelGGL((k<RLO, RHI>), dim3(256 * wg), dim3(512), 0, 0, d, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<RLO, RHI>), dim3(256 * wg), dim3(512), 0, 0, d, ITER);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s wg/CU=%d  %.1f ns per iteration of each wave-pair slot (x%d slots per SIMD)\n", name, wg, ms * 1e6 / ITER, wg);
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 3 * 512 * sizeof(float));
    run<1, 0>("MFMA waves only (1 per SIMD per WG)", d);
    run<0, 2>("VALU waves only (1 per SIMD per WG)", d);
    run<1, 2>("MFMA wave + VALU wave on each SIMD", d);
    run<1, 1>("MFMA + MFMA", d);
    run<2, 2>("VALU + VALU", d);
    run<3, 0>("dependent MFMA->tree (1 wave per SIMD per WG)", d);
    run<3, 3>("dependent MFMA->tree x2", d);
    run<4, 0>("pipelined MFMA | tree (1 wave)", d);
    run<4, 4>("pipelined MFMA | tree x2", d);
    return 0;
}
