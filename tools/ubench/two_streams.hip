// Do two kernels on two HIP streams overlap on this box?  Each kernel is `wgs` workgroups that spin for ~100 us.
// Prints the time of A alone, and of A and B issued to two streams, for a few grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t e0, e1, f;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&f, hipEventDisableTiming));
    const long long ticks = 10000;                        // 100 MHz counter: 100 us
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(unset)");
    for (int wgs : {1, 256, 2048, 8192}) {
        for (int both = 0; both < 2; ++both) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, a));
                if (both) { CK(hipEventRecord(f, a)); CK(hipStreamWaitEvent(b, f, 0)); }
                hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, a, ticks, (int*)nullptr);
                if (both) {
                    hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, b, ticks, (int*)nullptr);
                    CK(hipEventRecord(f, b)); CK(hipStreamWaitEvent(a, f, 0));
                }
                CK(hipEventRecord(e1, a));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%5d workgroups x 64 lanes, %s: %.1f us\n", wgs, both ? "A on stream 1 + B on stream 2" : "A alone                      ", best * 1e3f);
        }
    }
    return 0;
}
