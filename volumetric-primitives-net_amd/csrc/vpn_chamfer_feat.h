// vpn_chamfer_feat.h — the target-side features of the matrix-pipe Chamfer filter (fp32 planes + rows of 16-bit
// pieces + per-slice max norms) and the workgroup body that writes them.  Shared by chamfer.hip (chamfer_feat_kernel:
// both clouds of a Chamfer call) and sampler.hip (the forward launch of the training step writes the features of the
// cloud it has just sampled, and of the ground-truth cloud, so the feature kernel disappears from the step).
// Everything here is exact or covered by the filter's error bound whatever the translation unit's contraction mode.
#pragma once
#include "vpn_common.h"

namespace vpn {

constexpr int CM_ROWB = 48;              // bytes per bf16 row: K slots 0..15 for v_mfma_f32_32x32x16_bf16, 16..23 for v_mfma_f32_32x32x8_bf16
                                         // (21 used).  Also the LDS stride: 16 consecutive rows x 16 B hit 64 distinct banks (12 r mod 64)
// exact 3-way split of an fp32 into bf16 pieces (truncation): x == p0 + p1 + p2
__device__ inline void split3_bf16(float x, unsigned short p[3]) {
#pragma clang fp contract(off)
    const unsigned u0 = __float_as_uint(x);
    p[0] = (unsigned short)(u0 >> 16);
    const float r1 = x - __uint_as_float(u0 & 0xFFFF0000u);
    const unsigned u1 = __float_as_uint(r1);
    p[1] = (unsigned short)(u1 >> 16);
    const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
    p[2] = (unsigned short)(__float_as_uint(r2) >> 16);
}

constexpr float CM_S16 = 2048.0f;                            // coordinate scale of the fp16 rows (2^11)
constexpr int CM_ROWB16 = 32;            // bytes per fp16 row in HBM (16 K slots); LDS stride stays 48 B (bank-conflict free)
// X = h[0] + h[1] + r with |r| <= 2^-22 |X| + 2^-25 (fp16 pieces, round to nearest)
__device__ inline void split2_f16(float X, _Float16 h[2]) {
    h[0] = (_Float16)X;
    h[1] = (_Float16)(X - (float)h[0]);
}

// one 32-byte row: K slots  x (b1 b1 b2), y (...), z (...), S^2 |p|^2 as p1 2^15 + p2 2^4 + p3, 4 zeros
__device__ inline void write_row16(unsigned short* H, size_t row, float x, float y, float z, float n) {
#pragma clang fp contract(off)
    _Float16 hx[2], hy[2], hz[2], pn[3];
    split2_f16(CM_S16 * x, hx); split2_f16(CM_S16 * y, hy); split2_f16(CM_S16 * z, hz);
    if (n < 1.0e30f) {
        const float ns = (CM_S16 * CM_S16) * n;
        pn[0] = (_Float16)(ns * 3.0517578125e-05f);                     // 2^-15
        const float r1 = ns - (float)pn[0] * 32768.0f;
        pn[1] = (_Float16)(r1 * 0.0625f);                               // 2^-4
        pn[2] = (_Float16)(r1 - (float)pn[1] * 16.0f);
    } else {                                                            // padding sentinel: 65504 * 2^15 / S^2 = 512 > any t in range
        pn[0] = (_Float16)65504.0f; pn[1] = (_Float16)0.0f; pn[2] = (_Float16)0.0f;
    }
    auto bits = [](_Float16 v) { return (unsigned)__builtin_bit_cast(unsigned short, v); };
    auto pk = [&](_Float16 lo, _Float16 hi) { return bits(lo) | (bits(hi) << 16); };
    const _Float16 zero = (_Float16)0.0f;
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(H) + row * CM_ROWB16);
    dst[0] = make_uint4(pk(hx[0], hx[0]), pk(hx[1], hy[0]), pk(hy[0], hy[1]), pk(hz[0], hz[0]));
    dst[1] = make_uint4(pk(hz[1], pn[0]), pk(pn[1], pn[2]), pk(zero, zero), pk(zero, zero));
}

// feature planes F[b][4][Np] = (x, y, z, |p|^2) (padded with a never-winning sentinel) and
// H[b][Np][24] bf16 rows (48 B) for the bf16 filter: per coordinate the target pieces (b1 b1 b2 b1 b3 b2) that pair
// with the query pieces (a1 a2 a1 a3 a1 a2), then the three pieces of |p|^2 (paired with 1.0), then zeros.
// nmax[b][CFEAT_SLOTS]: max |p|^2 of each workgroup's slice (the filter takes the max of the slots: no atomics,
// no zero-initialised output, no workgroup that scans the whole cloud).
// one 48-byte row of bf16 pieces: K slots  x (b1 b1 b2 b1 b3 b2), y (...), z (...), |p|^2 (3 pieces), 3 zeros
__device__ inline void write_row(unsigned short* H, size_t row, float x, float y, float z, float n) {
#pragma clang fp contract(off)
    unsigned short px[3], py[3], pz[3], pn[3];
    split3_bf16(x, px); split3_bf16(y, py); split3_bf16(z, pz); split3_bf16(n, pn);
    auto pk = [](unsigned short lo, unsigned short hi) { return (unsigned)lo | ((unsigned)hi << 16); };
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(H) + row * CM_ROWB);
    dst[0] = make_uint4(pk(px[0], px[0]), pk(px[1], px[0]), pk(px[2], px[1]), pk(py[0], py[0]));
    dst[1] = make_uint4(pk(py[1], py[0]), pk(py[2], py[1]), pk(pz[0], pz[0]), pk(pz[1], pz[0]));
    dst[2] = make_uint4(pk(pz[2], pz[1]), pk(pn[0], pn[1]), pk(pn[2], 0), 0u);
}


constexpr int CFEAT_THREADS = 256;
#ifndef CFEAT_WGPTS
#define CFEAT_WGPTS 256             // points per feature workgroup (1024: r1; 256: 4x the workgroups, the kernel is latency-bound)
#endif
constexpr int CFEAT_SLOTS = 64;                    // slices (workgroups) per cloud and sample, at most: one lane each in the scan's epilogue
constexpr int CFEAT_PTS = 4;                       // points per lane in flight

struct FeatJob {          // one cloud: slices [0, ysplit) of a sample's workgroups belong to it
    const float* pts; int N, Np, ysplit;
    float* F; unsigned int* nmax; unsigned short* H;
    int rows16;           // rows as 32-byte fp16 pieces (PREC 2) instead of 48-byte bf16 pieces (PREC 1)
};

// workgroup id -> (sample, slice): sample b runs on XCD b / (B/8), where the filter kernel reads what is written
__device__ inline void feat_decode(int id, int B, int& b, int& sy) {
    const int per = B >> 3;
    if ((B & 7) == 0) { const int xcd = id & 7, r = id >> 3; b = xcd * per + r % per; sy = r / per; }
    else { b = id % B; sy = id / B; }
}

__device__ inline float feat_wave_max(float v) {      // values >= 0
#define VPN_FDPP(v, ctrl, rmask) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false))
    v = fmaxf(v, VPN_FDPP(v, 0x111, 0xf)); v = fmaxf(v, VPN_FDPP(v, 0x112, 0xf));
    v = fmaxf(v, VPN_FDPP(v, 0x114, 0xf)); v = fmaxf(v, VPN_FDPP(v, 0x118, 0xf));
    v = fmaxf(v, VPN_FDPP(v, 0x142, 0xa)); v = fmaxf(v, VPN_FDPP(v, 0x143, 0xc));
#undef VPN_FDPP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// planes and row of ONE point (j < Np; a padding point j >= N gets the never-winning sentinel); returns |p|^2 (0 for padding)
__device__ inline float feat_point(const FeatJob& J, int b, int j, float x, float y, float z) {
#pragma clang fp contract(off)
    float* f = J.F + (size_t)b * 4 * J.Np;
    float n = 3.0e38f, nv = 0.0f;
    if (j < J.N) { n = x * x + y * y + z * z; nv = n; } else { x = 0.f; y = 0.f; z = 0.f; }
    f[j] = x; f[J.Np + j] = y; f[2 * J.Np + j] = z; f[3 * J.Np + j] = n;
    if (J.H) {
        if (J.rows16) write_row16(J.H, (size_t)b * J.Np + j, x, y, z, n);
        else write_row(J.H, (size_t)b * J.Np + j, x, y, z, n);
    }
    return nv;
}

// per-workgroup bookkeeping of a slice: the max norm of the slice into its slot; slice 0 also zeroes the slots no
// slice owns.  `red`: CFEAT_THREADS / 64 floats of LDS.  Contains a barrier.
__device__ inline void feat_finish_slice(const FeatJob& J, int b, int by, float nv, float* red) {
    if (by == 0 && (int)threadIdx.x >= J.ysplit && threadIdx.x < CFEAT_SLOTS) J.nmax[b * CFEAT_SLOTS + threadIdx.x] = 0u;
    nv = feat_wave_max(nv);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < CFEAT_THREADS / 64; ++w) m = fmaxf(m, red[w]);
        J.nmax[b * CFEAT_SLOTS + by] = __float_as_uint(m);
    }
}

// slice `by` of sample b of the cloud J.pts (AoS xyz): one workgroup of CFEAT_THREADS lanes
__device__ inline void feat_slice(const FeatJob& J, int b, int by, float* red) {
    const float* pb = J.pts + (size_t)b * J.N * 3;
    float nv = 0.f;
    const int per = ((J.Np + J.ysplit - 1) / J.ysplit + 63) & ~63;
    const int jlo = by * per, jhi = min(J.Np, jlo + per);
    for (int j0p = jlo + threadIdx.x; j0p < jhi; j0p += CFEAT_PTS * CFEAT_THREADS) {
        float xs[CFEAT_PTS], ys[CFEAT_PTS], zs[CFEAT_PTS];
#pragma unroll
        for (int u = 0; u < CFEAT_PTS; ++u) {
            const int j = min(j0p + u * CFEAT_THREADS, J.N - 1);
            const F3 v3 = ld3(pb + j * 3);
            xs[u] = v3.x; ys[u] = v3.y; zs[u] = v3.z;
        }
#pragma unroll
        for (int u = 0; u < CFEAT_PTS; ++u) {
            const int j = j0p + u * CFEAT_THREADS;
            if (j >= jhi) break;
            nv = fmaxf(nv, feat_point(J, b, j, xs[u], ys[u], zs[u]));
        }
    }
    feat_finish_slice(J, b, by, nv, red);
}

// The same slice by the lanes [first, CFEAT_THREADS) of a workgroup and WITHOUT its barrier: the caller has one coming
// anyway (the sampler's pose barrier: the slice's loads then overlap the pose lane's chain instead of forming a phase of
// their own behind the sampling).  The waves' max norms go to red[wave]; after the caller's barrier ONE lane calls
// feat_slice_publish.  `first` is a multiple of 64.
__device__ inline void feat_slice_early(const FeatJob& J, int b, int by, int first, float* red) {
    if (by == 0 && (int)threadIdx.x >= J.ysplit && threadIdx.x < CFEAT_SLOTS) J.nmax[b * CFEAT_SLOTS + threadIdx.x] = 0u;
    if ((int)threadIdx.x < first) return;
    const float* pb = J.pts + (size_t)b * J.N * 3;
    float nv = 0.f;
    const int per = ((J.Np + J.ysplit - 1) / J.ysplit + 63) & ~63;
    const int jlo = by * per, jhi = min(J.Np, jlo + per), lanes = CFEAT_THREADS - first;
    for (int j = jlo + (int)threadIdx.x - first; j < jhi; j += lanes) {
        const F3 v3 = ld3(pb + min(j, J.N - 1) * 3);
        nv = fmaxf(nv, feat_point(J, b, j, v3.x, v3.y, v3.z));
    }
    nv = feat_wave_max(nv);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nv;
}
__device__ inline void feat_slice_publish(const FeatJob& J, int b, int by, int first, const float* red) {
    float m = red[first >> 6];
    for (int w = (first >> 6) + 1; w < CFEAT_THREADS / 64; ++w) m = fmaxf(m, red[w]);
    J.nmax[b * CFEAT_SLOTS + by] = __float_as_uint(m);
}

// host side (chamfer.hip): the two feature jobs of the fp16 filter inside a caller-provided Chamfer workspace, in the
// layout vpn_chamfer_fwd_ws(mode 7) expects.  p1 [B,N,3] is the cloud the caller is about to write (pts may be its
// address or null), p2 [B,M,3] the other one.  Returns 0, or VPN_E_BADARG if the workspace is short / misaligned or the
// automatic mode would not take the fp16 filter for these sizes.
int chamfer_feat_jobs(void* workspace, size_t workspace_bytes, int B, int N, int M, const float* p1, const float* p2,
                      FeatJob* job1, FeatJob* job2);
// the per-workgroup sums of the minima the fp16 scan leaves in that workspace: sums1 [B][*g1] (dist1), sums2 [B][*g2] (dist2)
int chamfer_wgsums(const void* workspace, size_t workspace_bytes, int B, int N, int M, const float** sums1, int* g1,
                   const float** sums2, int* g2);

}  // namespace vpn
