// vpn_common.h — device helpers shared by the gfx950 kernels (pose maths, Philox,
// wave64 reductions).  Written for CDNA4 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vpn_hip.h"

#define VPN_PI 3.1415927410125732f   // fp32 pi of modules/sampling/sphere.py:7, transform/rotate.py:4

#define VPN_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

namespace vpn {

// Optional per-kernel timing (bench.py): when enabled every launch is bracketed by HIP events on its stream.
void prof_begin(const char* name, hipStream_t s);
void prof_end(hipStream_t s);

}  // namespace vpn

#define VPN_LAUNCH(kern, grid, block, lds, strm, ...)                                  \
    do {                                                                               \
        vpn::prof_begin(#kern, strm);                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, strm, __VA_ARGS__);                 \
        vpn::prof_end(strm);                                                           \
    } while (0)

// the same under a given profile label (template instantiations of one kernel keep one name in the launch profile)
#define VPN_LAUNCH_AS(label, kern, grid, block, lds, strm, ...)                        \
    do {                                                                               \
        vpn::prof_begin(label, strm);                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, strm, __VA_ARGS__);                 \
        vpn::prof_end(strm);                                                           \
    } while (0)

namespace vpn {

struct Mat3 { float m[3][3]; };

struct Pose {          // everything derived from q (B,4) that fwd and bwd need
    Mat3 R;
    float x, y, z, w;  // unit quaternion
    float sh, ch;      // sin / cos of the half angle
    float inv_len;     // 1 / |(q012*sh, ch)|
};

// modules/transform/rotate.py:59-72 (refine_quaternions) + :28-46 (get_rotation_matrices)
__device__ inline Pose make_pose(float q0, float q1, float q2, float q3) {
    // every operation rounded by itself: the pose must come out the same bit for bit wherever this is inlined (the
    // sampler's pose lane, the record kernels, the backward), whatever the compiler would contract in that context
#pragma clang fp contract(off)
    Pose p;
    float r = q3 - floorf(q3);                 // torch `% 1` == remainder (sign of divisor)
    float h = ((r * 2.0f) * VPN_PI) / 2.0f;    // rotate.py:63
    p.sh = sinf(h);
    p.ch = cosf(h);
    float a = q0 * p.sh, b = q1 * p.sh, c = q2 * p.sh, d = p.ch;
    float len = sqrtf(a * a + b * b + c * c + d * d);
    p.inv_len = 1.0f / len;
    float x = a / len, y = b / len, z = c / len, w = d / len;
    p.x = x; p.y = y; p.z = z; p.w = w;
    float x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    float xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    p.R.m[0][0] = x2 - y2 - z2 + w2; p.R.m[0][1] = 2.0f * (xy - zw); p.R.m[0][2] = 2.0f * (xz + yw);
    p.R.m[1][0] = 2.0f * (xy + zw);  p.R.m[1][1] = -x2 + y2 - z2 + w2; p.R.m[1][2] = 2.0f * (yz - xw);
    p.R.m[2][0] = 2.0f * (xz - yw);  p.R.m[2][1] = 2.0f * (yz + xw);  p.R.m[2][2] = -x2 - y2 + z2 + w2;
    return p;
}

// Chain rule dL/dR (3x3, g[b][a] = dL/dR_ba) -> dL/dq (4), through the matrix
// entries, the normalisation and the half-angle map of rotate.py:59-72.
__device__ inline void pose_backward(const Pose& p, float q0, float q1, float q2,
                                     const float g[3][3], float gq[4]) {
    const float x = p.x, y = p.y, z = p.z, w = p.w;
    float gx = 2.0f * (x * g[0][0] + y * g[0][1] + z * g[0][2] + y * g[1][0] - x * g[1][1] - w * g[1][2]
                       + z * g[2][0] + w * g[2][1] - x * g[2][2]);
    float gy = 2.0f * (-y * g[0][0] + x * g[0][1] + w * g[0][2] + x * g[1][0] + y * g[1][1] + z * g[1][2]
                       - w * g[2][0] + z * g[2][1] - y * g[2][2]);
    float gz = 2.0f * (-z * g[0][0] - w * g[0][1] + x * g[0][2] + w * g[1][0] - z * g[1][1] + y * g[1][2]
                       + x * g[2][0] + y * g[2][1] + z * g[2][2]);
    float gw = 2.0f * (w * g[0][0] - z * g[0][1] + y * g[0][2] + z * g[1][0] + w * g[1][1] - x * g[1][2]
                       - y * g[2][0] + x * g[2][1] + w * g[2][2]);
    // r = rh / |rh|  ->  g_rh = (g_r - r (r.g_r)) / |rh|
    float dot = x * gx + y * gy + z * gz + w * gw;
    float ha = (gx - x * dot) * p.inv_len;
    float hb = (gy - y * dot) * p.inv_len;
    float hc = (gz - z * dot) * p.inv_len;
    float hd = (gw - w * dot) * p.inv_len;
    // rh = (q0 sh, q1 sh, q2 sh, ch),  h = (q3 mod 1) * pi
    gq[0] = ha * p.sh;
    gq[1] = hb * p.sh;
    gq[2] = hc * p.sh;
    float gh = (q0 * ha + q1 * hb + q2 * hc) * p.ch - hd * p.sh;
    gq[3] = gh * VPN_PI;
}

// ---- Philox4x32-10 (Salmon et al. SC'11); oracle: vpn_oracle.philox4x32_10
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform draws of point `p` of primitive `k` of GLOBAL sample `gb` (oracle: philox_uniforms)
__device__ inline void philox_uniform3(uint64_t seed, uint64_t gb, uint32_t k, uint32_t p, float u[3]) {
    uint32_t o[4];
    philox4x32_10(p, k, (uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    u[0] = (float)(o[0] >> 8) * 5.9604644775390625e-08f;   // 2^-24
    u[1] = (float)(o[1] >> 8) * 5.9604644775390625e-08f;
    u[2] = (float)(o[2] >> 8) * 5.9604644775390625e-08f;
}

// ---- xyz triples as ONE 12-byte access (global_load/store_dwordx3): three dword accesses of a gathered point are
// three L2 requests, and a kernel whose 2048 workgroups all gather at once is bound by the L2 request rate
struct F3 { float x, y, z; };
__device__ inline F3 ld3(const float* p) { return *reinterpret_cast<const F3*>(p); }
__device__ inline void st3(float* p, float x, float y, float z) { *reinterpret_cast<F3*>(p) = F3{x, y, z}; }

// ---- wave64 reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace vpn
