// chamfer.hip — Chamfer nearest-neighbour reduction (both directions), loss and
// backward scatter for gfx950.
//
// Replaces the dense B*N*M expression of ChamferDistanceLoss.forward
// (modules/loss/chamfer_distance.py:14-30) and its autograd graph.  Nothing of size
// B*N*M is materialised: each lane owns R query points in registers, the target cloud
// streams through an LDS tile (SoA, read as ds_read_b128 broadcasts), and only the N+M
// minima / arg-minima are written.
//
// Bit-exactness contract (north_star: "index-exact for the Chamfer argmin"):
//   d2 = ((dx*dx) + (dy*dy)) + (dz*dz) with every product and sum rounded separately
//   (no FMA contraction: the pragma below) == torch.sum(diff*diff, dim=3);
//   dist = correctly rounded sqrtf(d2);
//   argmin = lowest index among equal *sqrt* values, which is what torch.min(dim)
//   returns on sqrt(dist) (chamfer_distance.py:19-23).  sqrt can map two different d2
//   onto one float, so the scan on d2 also tracks `prev` = the running minimum just
//   before the last update (= min over all earlier indices); if sqrtf(prev) equals the
//   final minimum the lane re-scans the earlier indices with the sqrt compare.
#include "vpn_common.h"

#pragma clang fp contract(off)

namespace vpn {

constexpr int CH_BLOCK = 256;
constexpr int CH_TILE = 1024;   // targets per LDS tile: 3 * 4 KB

__device__ inline float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}

// queries q [B,Nq,3], targets t [B,Nt,3] -> dist [B,Nq], idx [B,Nq]
template <int R>
__global__ __launch_bounds__(CH_BLOCK) void chamfer_nn_kernel(const float* __restrict__ qpts,
                                                              const float* __restrict__ tpts, int Nq, int Nt,
                                                              float* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_idx) {
    __shared__ __attribute__((aligned(16))) float sx[CH_TILE];
    __shared__ __attribute__((aligned(16))) float sy[CH_TILE];
    __shared__ __attribute__((aligned(16))) float sz[CH_TILE];
    const int b = blockIdx.y;
    const float* qb = qpts + (size_t)b * Nq * 3;
    const float* tb = tpts + (size_t)b * Nt * 3;
    const int q0 = blockIdx.x * (CH_BLOCK * R) + threadIdx.x;

    float ax[R], ay[R], az[R], best[R], prev[R];
    int bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int qi = q0 + r * CH_BLOCK;
        int qc = qi < Nq ? qi : Nq - 1;   // clamp: out-of-range lanes compute a valid point and do not store
        ax[r] = qb[qc * 3]; ay[r] = qb[qc * 3 + 1]; az[r] = qb[qc * 3 + 2];
        best[r] = __builtin_inff(); prev[r] = __builtin_inff(); bidx[r] = 0;
    }

    for (int t0 = 0; t0 < Nt; t0 += CH_TILE) {
        const int cnt = min(CH_TILE, Nt - t0);
        const int cnt4 = (cnt + 3) & ~3;
        __syncthreads();
        // coalesced AoS read of the tile, SoA write into LDS
        for (int i = threadIdx.x; i < cnt * 3; i += CH_BLOCK) {
            float val = tb[(size_t)t0 * 3 + i];
            int p = i / 3, c = i - p * 3;
            float* dst = c == 0 ? sx : (c == 1 ? sy : sz);
            dst[p] = val;
        }
        // pad to a multiple of 4 with a far sentinel (d2 = +inf never beats a finite minimum)
        if (threadIdx.x < cnt4 - cnt) {
            sx[cnt + threadIdx.x] = 3.0e38f; sy[cnt + threadIdx.x] = 3.0e38f; sz[cnt + threadIdx.x] = 3.0e38f;
        }
        __syncthreads();
        for (int j = 0; j < cnt4; j += 4) {
            const float4 X = *reinterpret_cast<const float4*>(&sx[j]);
            const float4 Y = *reinterpret_cast<const float4*>(&sy[j]);
            const float4 Z = *reinterpret_cast<const float4*>(&sz[j]);
            const float xs[4] = {X.x, X.y, X.z, X.w};
            const float ys[4] = {Y.x, Y.y, Y.z, Y.w};
            const float zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float d2 = dist2_exact(ax[r], ay[r], az[r], xs[jj], ys[jj], zs[jj]);
                    bool up = d2 < best[r];
                    prev[r] = up ? best[r] : prev[r];
                    best[r] = up ? d2 : best[r];
                    bidx[r] = up ? (t0 + j + jj) : bidx[r];
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        int qi = q0 + r * CH_BLOCK;
        if (qi >= Nq) continue;
        float s = sqrtf(best[r]);
        int idx = bidx[r];
        // rare: an earlier target has a larger d2 that rounds to the same sqrt -> it wins the tie
        if (sqrtf(prev[r]) == s) {
            for (int j = 0; j < idx; ++j) {
                float d2 = dist2_exact(ax[r], ay[r], az[r], tb[j * 3], tb[j * 3 + 1], tb[j * 3 + 2]);
                if (sqrtf(d2) == s) { idx = j; break; }
            }
        }
        out_dist[(size_t)b * Nq + qi] = s;
        out_idx[(size_t)b * Nq + qi] = idx;
    }
}

// loss_b = w1 * mean(dist1[b,:]) + w2 * mean(dist2[b,:])      (chamfer_distance.py:25-28)
__global__ __launch_bounds__(256) void chamfer_loss_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                           int N, int M, float w1, float w2,
                                                           float* __restrict__ loss_b) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) s1 += d1[(size_t)b * N + i];
    for (int j = threadIdx.x; j < M; j += 256) s2 += d2[(size_t)b * M + j];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        float c = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        loss_b[b] = w1 * (a / (float)N) + w2 * (c / (float)M);
    }
}

// direct terms: every point's own nearest neighbour.  grad_p1[i] = g1 (a_i - b_j*)/d ;
// grad_p2[j] = -g2 (a_i* - b_j)/d.  Written, not accumulated.
__global__ __launch_bounds__(256) void chamfer_bwd_direct_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ d1,
    const int32_t* __restrict__ i1, const float* __restrict__ d2, const int32_t* __restrict__ i2,
    const float* __restrict__ gl, int N, int M, float w1, float w2, float* __restrict__ g1,
    float* __restrict__ g2) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float g = gl[b];
    if (g1 && e < N) {
        const float* a = p1 + ((size_t)b * N + e) * 3;
        const float* c = p2 + ((size_t)b * M + i1[(size_t)b * N + e]) * 3;
        float coef = (g * w1 / (float)N) / d1[(size_t)b * N + e];
        float* o = g1 + ((size_t)b * N + e) * 3;
        o[0] = coef * (a[0] - c[0]); o[1] = coef * (a[1] - c[1]); o[2] = coef * (a[2] - c[2]);
    }
    if (g2 && e < M) {
        const float* c = p2 + ((size_t)b * M + e) * 3;
        const float* a = p1 + ((size_t)b * N + i2[(size_t)b * M + e]) * 3;
        float coef = (g * w2 / (float)M) / d2[(size_t)b * M + e];
        float* o = g2 + ((size_t)b * M + e) * 3;
        o[0] = coef * (c[0] - a[0]); o[1] = coef * (c[1] - a[1]); o[2] = coef * (c[2] - a[2]);
    }
}

// scatter terms: the gradient a point receives for being somebody else's nearest neighbour.
__global__ __launch_bounds__(256) void chamfer_bwd_scatter_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ d1,
    const int32_t* __restrict__ i1, const float* __restrict__ d2, const int32_t* __restrict__ i2,
    const float* __restrict__ gl, int N, int M, float w1, float w2, float* __restrict__ g1,
    float* __restrict__ g2) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float g = gl[b];
    if (g2 && e < N) {   // direction 1 pair (e, i1[e]) pushes on p2[i1[e]]
        const int j = i1[(size_t)b * N + e];
        const float* a = p1 + ((size_t)b * N + e) * 3;
        const float* c = p2 + ((size_t)b * M + j) * 3;
        float coef = (g * w1 / (float)N) / d1[(size_t)b * N + e];
        float* o = g2 + ((size_t)b * M + j) * 3;
        atomicAdd(o + 0, coef * (c[0] - a[0]));
        atomicAdd(o + 1, coef * (c[1] - a[1]));
        atomicAdd(o + 2, coef * (c[2] - a[2]));
    }
    if (g1 && e < M) {   // direction 2 pair (i2[e], e) pushes on p1[i2[e]]
        const int i = i2[(size_t)b * M + e];
        const float* c = p2 + ((size_t)b * M + e) * 3;
        const float* a = p1 + ((size_t)b * N + i) * 3;
        float coef = (g * w2 / (float)M) / d2[(size_t)b * M + e];
        float* o = g1 + ((size_t)b * N + i) * 3;
        atomicAdd(o + 0, coef * (a[0] - c[0]));
        atomicAdd(o + 1, coef * (a[1] - c[1]));
        atomicAdd(o + 2, coef * (a[2] - c[2]));
    }
}

template <int R>
static int launch_nn(const float* q, const float* t, int B, int Nq, int Nt, float* d, int32_t* idx, hipStream_t s) {
    int gx = (Nq + CH_BLOCK * R - 1) / (CH_BLOCK * R);
    hipLaunchKernelGGL(chamfer_nn_kernel<R>, dim3(gx, B), dim3(CH_BLOCK), 0, s, q, t, Nq, Nt, d, idx);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// pick queries-per-lane so that the launch still has >= ~4 waves per SIMD of the 256 CUs
static int pick_r(int B, int Nq) {
    long waves4 = ((long)B * Nq + 255) / 256;   // waves at R = 4
    if (waves4 >= 4096) return 4;
    if (waves4 * 2 >= 4096) return 2;
    return 1;
}

static int nn_dispatch(const float* q, const float* t, int B, int Nq, int Nt, float* d, int32_t* idx, hipStream_t s) {
    switch (pick_r(B, Nq)) {
        case 4: return launch_nn<4>(q, t, B, Nq, Nt, d, idx, s);
        case 2: return launch_nn<2>(q, t, B, Nq, Nt, d, idx, s);
        default: return launch_nn<1>(q, t, B, Nq, Nt, d, idx, s);
    }
}

}  // namespace vpn

using namespace vpn;

extern "C" int vpn_chamfer_nn(const float* queries, const float* targets, int B, int Nq, int Nt, float* dist,
                              int32_t* idx, void* stream) {
    if (!queries || !targets || !dist || !idx) return VPN_E_BADARG;
    if (B <= 0 || Nq <= 0 || Nt <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    return nn_dispatch(queries, targets, B, Nq, Nt, dist, idx, (hipStream_t)stream);
}

extern "C" int vpn_chamfer_fwd(const float* p1, const float* p2, int B, int N, int M, float* dist1, int32_t* idx1,
                               float* dist2, int32_t* idx2, void* stream) {
    if (!p1 || !p2 || !dist1 || !idx1 || !dist2 || !idx2) return VPN_E_BADARG;
    if (B <= 0 || N <= 0 || M <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    int rc = nn_dispatch(p1, p2, B, N, M, dist1, idx1, (hipStream_t)stream);
    if (rc) return rc;
    return nn_dispatch(p2, p1, B, M, N, dist2, idx2, (hipStream_t)stream);
}

extern "C" int vpn_chamfer_loss(const float* dist1, const float* dist2, int B, int N, int M, float w1, float w2,
                                float* loss_b, void* stream) {
    if (!dist1 || !dist2 || !loss_b) return VPN_E_BADARG;
    if (B <= 0 || N <= 0 || M <= 0) return VPN_E_BADARG;
    hipLaunchKernelGGL(chamfer_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dist1, dist2, N, M, w1, w2,
                       loss_b);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_chamfer_bwd(const float* p1, const float* p2, const float* dist1, const int32_t* idx1,
                               const float* dist2, const int32_t* idx2, const float* grad_loss_b, int B, int N,
                               int M, float w1, float w2, float* grad_p1, float* grad_p2, void* stream) {
    if (!p1 || !p2 || !dist1 || !idx1 || !dist2 || !idx2 || !grad_loss_b) return VPN_E_BADARG;
    if (B <= 0 || N <= 0 || M <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    if (!grad_p1 && !grad_p2) return 0;
    int mx = N > M ? N : M;
    dim3 grid((mx + 255) / 256, B);
    hipLaunchKernelGGL(chamfer_bwd_direct_kernel, grid, dim3(256), 0, (hipStream_t)stream, p1, p2, dist1, idx1,
                       dist2, idx2, grad_loss_b, N, M, w1, w2, grad_p1, grad_p2);
    VPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(chamfer_bwd_scatter_kernel, grid, dim3(256), 0, (hipStream_t)stream, p1, p2, dist1, idx1,
                       dist2, idx2, grad_loss_b, N, M, w1, w2, grad_p1, grad_p2);
    VPN_LAUNCH_CHECK();
    return 0;
}
