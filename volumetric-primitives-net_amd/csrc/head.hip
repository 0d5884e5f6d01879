// head.hip — post-processing of the network heads into the packed primitive parameters (row f4).
//
// Replaces restrict_range + split + restrict_volumes of the reference's model
// (modules/network/vpnet_one_resnet.py:34-41, :67-85; same code in vpnet_two_resnet.py and sdnet.py):
//   volumes    [B,3K] -> sigmoid(x) + 0.1            (or clamp(x, cmin + 1e-8, cmax)),  then / VOLUME_RESTRICT[j]
//   rotates    [B,4K] -> sigmoid(x)                   (or clamp(x, -1, 1))
//   translates [B,3K] -> tanh(x)                      (or clamp(x, -1, 1))
// and the K python lists of (B,3|4|3) tensors that train.py carries around become the packed [B,K,10] rows
// (v0 v1 v2 q0 q1 q2 q3 t0 t1 t2) that the sampler, the raster and their backward kernels read directly.
// (restrict_volumes writes in place into views of a split, which current PyTorch refuses under autograd —
// SURVEY.md Appendix C; the fused op has no such problem.)
#include "vpn_common.h"

namespace vpn {

__device__ inline float sigmoidf(float x) {                     // overflow-free on both sides
    const float e = __expf(-fabsf(x));
    const float s = 1.0f / (1.0f + e);
    return x >= 0.0f ? s : e * s;
}

// one thread per packed element; BACKWARD: out = dL/draw given g = dL/dparams
template <bool BACKWARD>
__global__ __launch_bounds__(256) void head_pack_kernel(const float* __restrict__ volumes,
                                                        const float* __restrict__ rotates,
                                                        const float* __restrict__ translates,
                                                        const float* __restrict__ grad_params, int K, int total,
                                                        int is_sigmoid, float cmin, float cmax, float r0, float r1,
                                                        float r2, float* __restrict__ params,
                                                        float* __restrict__ grad_volumes,
                                                        float* __restrict__ grad_rotates,
                                                        float* __restrict__ grad_translates) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int f = e % VPN_PARAM_STRIDE, bk = e / VPN_PARAM_STRIDE, b = bk / K, k = bk - b * K;
    // source tensor and index of this field: split(3|4|3, dim=1) takes consecutive column groups (:36-38)
    const float* src; float* gdst; int idx;
    if (f < 3) { src = volumes; gdst = grad_volumes; idx = b * 3 * K + k * 3 + f; }
    else if (f < 7) { src = rotates; gdst = grad_rotates; idx = b * 4 * K + k * 4 + (f - 3); }
    else { src = translates; gdst = grad_translates; idx = b * 3 * K + k * 3 + (f - 7); }
    const float x = src[idx];
    float y, dy;
    if (is_sigmoid) {
        if (f < 7) { const float s = sigmoidf(x); y = f < 3 ? s + 0.1f : s; dy = s * (1.0f - s); }     // :70-71
        else { y = tanhf(x); dy = 1.0f - y * y; }                                                        // :72
    } else {
        const float lo = f < 3 ? cmin + 1e-8f : -1.0f, hi = f < 3 ? cmax : 1.0f;                          // :74-76
        y = fminf(fmaxf(x, lo), hi);
        dy = (x >= lo && x <= hi) ? 1.0f : 0.0f;                 // torch.clamp passes the gradient on the closed interval
    }
    if (f < 3) { const float r = f == 0 ? r0 : (f == 1 ? r1 : r2); y = y / r; dy = dy / r; }              // :81-84
    if (BACKWARD) { if (gdst) gdst[idx] = grad_params[e] * dy; }
    else params[e] = y;
}

}  // namespace vpn

using namespace vpn;

static int head_check(const void* v, const void* q, const void* t, int B, int K, float r0, float r1, float r2) {
    if (!v || !q || !t || B <= 0 || K <= 0) return VPN_E_BADARG;
    if (!(r0 > 0.0f) || !(r1 > 0.0f) || !(r2 > 0.0f)) return VPN_E_BADARG;
    if ((long long)B * K * VPN_PARAM_STRIDE > 0x7fffffffLL) return VPN_E_TOOBIG;
    return 0;
}

extern "C" int vpn_head_pack_fwd(const float* volumes, const float* rotates, const float* translates, int B, int K,
                                 int is_sigmoid, float clamp_min, float clamp_max, float restrict0, float restrict1,
                                 float restrict2, float* params, void* stream) {
    int rc = head_check(volumes, rotates, translates, B, K, restrict0, restrict1, restrict2);
    if (rc) return rc;
    if (!params) return VPN_E_BADARG;
    const int total = B * K * VPN_PARAM_STRIDE;
    VPN_LAUNCH(head_pack_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, volumes, rotates,
               translates, (const float*)nullptr, K, total, is_sigmoid, clamp_min, clamp_max, restrict0, restrict1,
               restrict2, params, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_head_pack_bwd(const float* volumes, const float* rotates, const float* translates,
                                 const float* grad_params, int B, int K, int is_sigmoid, float clamp_min,
                                 float clamp_max, float restrict0, float restrict1, float restrict2,
                                 float* grad_volumes, float* grad_rotates, float* grad_translates, void* stream) {
    int rc = head_check(volumes, rotates, translates, B, K, restrict0, restrict1, restrict2);
    if (rc) return rc;
    if (!grad_params) return VPN_E_BADARG;
    const int total = B * K * VPN_PARAM_STRIDE;
    VPN_LAUNCH(head_pack_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, volumes, rotates,
               translates, grad_params, K, total, is_sigmoid, clamp_min, clamp_max, restrict0, restrict1, restrict2,
               (float*)nullptr, grad_volumes, grad_rotates, grad_translates);
    VPN_LAUNCH_CHECK();
    return 0;
}
