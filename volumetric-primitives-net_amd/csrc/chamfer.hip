// chamfer.hip — Chamfer nearest-neighbour reduction (both directions), loss and
// backward scatter for gfx950.
//
// Replaces the dense B*N*M expression of ChamferDistanceLoss.forward
// (modules/loss/chamfer_distance.py:14-30) and its autograd graph.  Nothing of size
// B*N*M is materialised: each lane owns R query points in registers, the target cloud
// streams through wave-private LDS tiles (SoA, read as ds_read_b128 broadcasts), two targets
// are evaluated per packed-fp32 instruction, and only the N+M minima / arg-minima are written.
//
// Bit-exactness contract (north_star: "index-exact for the Chamfer argmin"):
//   d2 = ((dx*dx) + (dy*dy)) + (dz*dz) with every product and sum rounded separately
//   (no FMA contraction: the pragma below) == torch.sum(diff*diff, dim=3);
//   dist = correctly rounded sqrtf(d2);
//   argmin = lowest index among equal *sqrt* values, which is what torch.min(dim)
//   returns on sqrt(dist) (chamfer_distance.py:19-23).  sqrt can map two different d2
//   onto one float, so the scan on d2 also tracks the running minimum just before the
//   last update (= min over all targets of earlier groups); if its sqrt equals the final
//   minimum the wave re-scans the earlier targets with the sqrt compare.
#include "vpn_common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace vpn {

constexpr int CH_BLOCK = 256;    // 4 waves: same queries, each wave scans one contiguous quarter of the targets
constexpr int CH_WTILE = 256;    // targets per wave-private LDS tile (SoA, 3 KB)

typedef float f2 __attribute__((ext_vector_type(2)));

#ifdef VPN_CHAMFER_DEBUG
__device__ unsigned long long g_dbg[8];
#endif

__device__ inline float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}

// queries q [B,Nq,3], targets t [B,Nt,3] -> dist [B,Nq], idx [B,Nq].
// Block = 64*R queries (lane owns R of them in registers) x 4 waves; wave w scans targets
// [w*Nw, (w+1)*Nw) through its own LDS tile (no block barrier in the scan).
// Per group of 8 targets: 8 d2 (two per packed instruction), v_min3 tree, ONE compare and three
// selects: the scan tracks (best d2, base index of the group that last lowered it, best before
// that group).  The exact index inside the group and the sqrt tie rule are resolved in the
// epilogue.  The four partial results are merged in target order.
template <int R>
__global__ __launch_bounds__(CH_BLOCK) void chamfer_nn_kernel(const float* __restrict__ qpts,
                                                              const float* __restrict__ tpts, int Nq, int Nt,
                                                              float* __restrict__ out_dist,
                                                              int32_t* __restrict__ out_idx) {
    __shared__ __attribute__((aligned(16))) float lds[4][3][CH_WTILE];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* qb = qpts + (size_t)b * Nq * 3;
    const float* tb = tpts + (size_t)b * Nt * 3;
    const int q0 = blockIdx.x * (64 * R) + lane;

    float ax[R], ay[R], az[R], best[R], gprev[R];
    int gidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int qi = q0 + r * 64;
        int qc = qi < Nq ? qi : Nq - 1;   // clamp: out-of-range lanes compute a valid point and do not store
        ax[r] = qb[qc * 3]; ay[r] = qb[qc * 3 + 1]; az[r] = qb[qc * 3 + 2];
        best[r] = __builtin_inff(); gprev[r] = __builtin_inff(); gidx[r] = 0;
    }

    const int Nw = (((Nt + 3) >> 2) + 7) & ~7;              // per-wave share, multiple of 8
    const int w_lo = min(wave * Nw, Nt), w_hi = min(w_lo + Nw, Nt);
    float* sx = lds[wave][0];
    float* sy = lds[wave][1];
    float* sz = lds[wave][2];
    for (int t0 = w_lo; t0 < w_hi; t0 += CH_WTILE) {
        const int cnt = min(CH_WTILE, w_hi - t0);
        const int cnt8 = (cnt + 7) & ~7;
        // the wave's previous tile must be fully read before it is overwritten (LDS ops of one
        // wave execute in order; the fence only stops the compiler from reordering them)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = lane; i < cnt * 3; i += 64) {          // coalesced AoS read, SoA write
            float val = tb[(size_t)t0 * 3 + i];
            int p = i / 3, c = i - p * 3;
            lds[wave][c][p] = val;
        }
        if (lane < cnt8 - cnt) {   // pad to a multiple of 8 with a far sentinel (d2 = +inf never wins)
            sx[cnt + lane] = 3.0e38f; sy[cnt + lane] = 3.0e38f; sz[cnt + lane] = 3.0e38f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int j = 0; j < cnt8; j += 8) {
            const float4 Xa = *reinterpret_cast<const float4*>(&sx[j]), Xb = *reinterpret_cast<const float4*>(&sx[j + 4]);
            const float4 Ya = *reinterpret_cast<const float4*>(&sy[j]), Yb = *reinterpret_cast<const float4*>(&sy[j + 4]);
            const float4 Za = *reinterpret_cast<const float4*>(&sz[j]), Zb = *reinterpret_cast<const float4*>(&sz[j + 4]);
            const f2 X[4] = {{Xa.x, Xa.y}, {Xa.z, Xa.w}, {Xb.x, Xb.y}, {Xb.z, Xb.w}};
            const f2 Y[4] = {{Ya.x, Ya.y}, {Ya.z, Ya.w}, {Yb.x, Yb.y}, {Yb.z, Yb.w}};
            const f2 Z[4] = {{Za.x, Za.y}, {Za.z, Za.w}, {Zb.x, Zb.y}, {Zb.z, Zb.w}};
            const int jb = t0 + j;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const f2 qx = {ax[r], ax[r]}, qy = {ay[r], ay[r]}, qz = {az[r], az[r]};
                f2 d[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {   // every product and sum rounded separately (contract off)
                    const f2 dx = qx - X[h], dy = qy - Y[h], dz = qz - Z[h];
                    d[h] = ((dx * dx) + (dy * dy)) + (dz * dz);
                }
                float m = __builtin_fminf(__builtin_fminf(d[0].x, d[0].y), d[1].x);
                m = __builtin_fminf(__builtin_fminf(m, d[1].y), d[2].x);
                m = __builtin_fminf(__builtin_fminf(m, d[2].y), d[3].x);
                m = __builtin_fminf(m, d[3].y);
                const bool up = m < best[r];            // strict: an equal d2 in a later group never wins
                gprev[r] = up ? best[r] : gprev[r];
                best[r] = up ? m : best[r];
                gidx[r] = up ? jb : gidx[r];
            }
        }
    }

    // ordered merge of the four target quarters through LDS (reuses the tile storage)
    __syncthreads();
    float* mb = &lds[0][0][0];                      // [3 waves][3 values][R][64]
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float* dst = mb + (((wave - 1) * 3) * R + r) * 64 + lane;
            dst[0] = best[r];
            dst[R * 64] = __int_as_float(gidx[r]);
            dst[2 * R * 64] = gprev[r];
        }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* src = mb + ((w * 3) * R + r) * 64 + lane;
            const float b2 = src[0], p2 = src[2 * R * 64];
            const int i2 = __float_as_int(src[R * 64]);
            const bool up = b2 < best[r];           // later quarter wins only if strictly smaller
            gprev[r] = up ? fminf(best[r], p2) : gprev[r];   // everything in earlier quarters precedes its group
            best[r] = up ? b2 : best[r];
            gidx[r] = up ? i2 : gidx[r];
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int qi = q0 + r * 64;
        const float s = sqrtf(best[r]);
        // inside the winning group: the first target whose sqrt equals s (this covers both the
        // first target attaining best and an earlier in-group target that ties after sqrt)
        const int g = gidx[r];
        int idx = g;
#pragma unroll
        for (int e = 7; e >= 0; --e) {
            const int j = min(g + e, Nt - 1);
            const float d2 = dist2_exact(ax[r], ay[r], az[r], tb[j * 3], tb[j * 3 + 1], tb[j * 3 + 2]);
            if (g + e < Nt && sqrtf(d2) == s) idx = g + e;
        }
#ifndef VPN_CHAMFER_NO_RESCAN
        // Rare (about one query in 1e5..1e6): a target of an EARLIER group has a larger d2 that
        // rounds to the same sqrt, so it wins the reference's tie rule.  gprev = min d2 over all
        // earlier groups; the whole wave re-scans them for that one query, 64 per step, and
        // stops at the first step with a hit.
        unsigned long long need = __ballot(qi < Nq && sqrtf(gprev[r]) == s);
        while (need) {
            const int src = __builtin_ctzll(need);
            need &= need - 1;
            const float qx = __shfl(ax[r], src, 64), qy = __shfl(ay[r], src, 64), qz = __shfl(az[r], src, 64);
            const float ss = __shfl(s, src, 64);
            const int lim = __shfl(g, src, 64);
            int found = -1;
            for (int base = 0; base < lim; base += 64) {
                const int j = base + lane;
                bool hit = false;
                if (j < lim) hit = sqrtf(dist2_exact(qx, qy, qz, tb[j * 3], tb[j * 3 + 1], tb[j * 3 + 2])) == ss;
                const unsigned long long hm = __ballot(hit);
                if (hm) { found = base + __builtin_ctzll(hm); break; }
            }
            if (lane == src && found >= 0) idx = found;
#ifdef VPN_CHAMFER_DEBUG
            if (lane == src) { atomicAdd(&g_dbg[0], 1ull); atomicAdd(&g_dbg[3], (unsigned long long)lim); }
#endif
        }
#endif
        if (qi < Nq) {
            out_dist[(size_t)b * Nq + qi] = s;
            out_idx[(size_t)b * Nq + qi] = idx;
        }
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
    int gx = (Nq + 64 * R - 1) / (64 * R);
    hipLaunchKernelGGL(chamfer_nn_kernel<R>, dim3(gx, B), dim3(CH_BLOCK), 0, s, q, t, Nq, Nt, d, idx);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// queries per lane: 4 when that still leaves >= 8 waves per SIMD in flight, else 2
static int pick_r(int B, int Nq) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("VPN_CHAMFER_R");      // tuning override
        forced = e ? atoi(e) : 0;
    }
    if (forced == 2 || forced == 4) return forced;
    long blocks4 = ((long)B * Nq + 255) / 256;
    return blocks4 >= 2048 ? 4 : 2;
}

static int nn_dispatch(const float* q, const float* t, int B, int Nq, int Nt, float* d, int32_t* idx, hipStream_t s) {
    return pick_r(B, Nq) == 4 ? launch_nn<4>(q, t, B, Nq, Nt, d, idx, s) : launch_nn<2>(q, t, B, Nq, Nt, d, idx, s);
}

}  // namespace vpn

using namespace vpn;

#ifdef VPN_CHAMFER_DEBUG
extern "C" int vpn_debug_read(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dbg), 64);
}
#endif

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
