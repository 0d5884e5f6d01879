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
#include "vpn_chamfer_feat.h"
#include "vpn_raster_common.h"      // the tile-order rider of the training step (raster_order_wg)
#include <stdlib.h>
#include <string.h>

#pragma clang fp contract(off)

namespace vpn {

constexpr int CH_BLOCK = 256;    // 4 waves: same queries, each wave scans one contiguous quarter of the targets
constexpr int CH_WTILE = 256;    // targets per wave-private LDS tile (SoA, 3 KB)

typedef float f2 __attribute__((ext_vector_type(2)));

#ifdef CM_EXP_TRACE
// timeline experiment (tools/scan_timeline.py): per scan workgroup start, end of the tile loop, end of the per-query
// finish, end (after its fix-up) on the 100 MHz wall clock, job (direction) and the number of undecided queries it resolved
__device__ unsigned long long g_ctrace[8192 * 8];
extern "C" int vpn_debug_scan_trace(void* dst, int nwg) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ctrace), (size_t)nwg * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToDevice);
}
#endif
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
        if (qi < Nq) {
            out_dist[(size_t)b * Nq + qi] = s;
            out_idx[(size_t)b * Nq + qi] = idx;
        }
    }
}

// loss_b = w1 * mean(dist1[b,:]) + w2 * mean(dist2[b,:])      (chamfer_distance.py:25-28)
// sum of n floats by one workgroup of 256 lanes: float4 loads when the row is 16-byte aligned, 4 independent
// accumulators so the loads pipeline; fixed summation order (deterministic)
__device__ inline float row_sum_256(const float* __restrict__ p, int n) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) { const float4 v = p4[i]; a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w; }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += 256) a0 += p[i];
    return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void chamfer_loss_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                           int N, int M, float w1, float w2,
                                                           float* __restrict__ loss_b) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    float s1 = wave_sum(row_sum_256(d1 + (size_t)b * N, N));
    float s2 = wave_sum(row_sum_256(d2 + (size_t)b * M, M));
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

// Backward for one sample per workgroup when a cloud's gradient fits in LDS (N*12 B <= 144 KB): the
// direct term initialises an LDS copy of the gradient, the scatter term (many GT points pushing on the
// same predicted point) is accumulated with LDS atomics, and the result leaves with coalesced stores.
// Replaces chamfer_bwd_direct + chamfer_bwd_scatter (global float atomics to random rows: 26 us at C3).
constexpr int CB_THREADS = 1024;
constexpr int CB_MAX_POINTS = 12288;

// grad of cloud A [Na] (queries of direction dA: dA/iA over Na, scattered into by direction dB/iB over Nb)
__global__ __launch_bounds__(CB_THREADS) void chamfer_bwd_lds_kernel(
    const float* __restrict__ pa, const float* __restrict__ pb, const float* __restrict__ dA,
    const int32_t* __restrict__ iA, const float* __restrict__ dB, const int32_t* __restrict__ iB,
    const float* __restrict__ gl, int Na, int Nb, float wA, float wB, float* __restrict__ ga) {
    extern __shared__ float acc[];   // [Na*3]
    const int b = blockIdx.x;
    const float g = gl[b];
    const float* A = pa + (size_t)b * Na * 3;
    const float* Bp = pb + (size_t)b * Nb * 3;
    const float ca = g * wA / (float)Na, cb = g * wB / (float)Nb;
    for (int e = threadIdx.x; e < Na; e += CB_THREADS) {          // own nearest neighbour
        const int j = iA[(size_t)b * Na + e];
        const float coef = ca / dA[(size_t)b * Na + e];
        acc[e * 3 + 0] = coef * (A[e * 3 + 0] - Bp[j * 3 + 0]);
        acc[e * 3 + 1] = coef * (A[e * 3 + 1] - Bp[j * 3 + 1]);
        acc[e * 3 + 2] = coef * (A[e * 3 + 2] - Bp[j * 3 + 2]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < Nb; e += CB_THREADS) {          // being somebody else's nearest neighbour
        const int i = iB[(size_t)b * Nb + e];
        const float coef = cb / dB[(size_t)b * Nb + e];
        atomicAdd(&acc[i * 3 + 0], coef * (A[i * 3 + 0] - Bp[e * 3 + 0]));
        atomicAdd(&acc[i * 3 + 1], coef * (A[i * 3 + 1] - Bp[e * 3 + 1]));
        atomicAdd(&acc[i * 3 + 2], coef * (A[i * 3 + 2] - Bp[e * 3 + 2]));
    }
    __syncthreads();
    float* out = ga + (size_t)b * Na * 3;
    for (int i = threadIdx.x; i < Na * 3; i += CB_THREADS) out[i] = acc[i];
}

static int launch_bwd_lds(const float* pa, const float* pb, const float* dA, const int32_t* iA, const float* dB,
                          const int32_t* iB, const float* gl, int B, int Na, int Nb, float wA, float wB, float* ga,
                          hipStream_t s) {
    static int raised = 0;
    const size_t lds = (size_t)Na * 3 * sizeof(float);
    if (lds > 65536 && !raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chamfer_bwd_lds_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, CB_MAX_POINTS * 12);
        if (e != hipSuccess) return (int)e;
        raised = 1;
    }
    VPN_LAUNCH(chamfer_bwd_lds_kernel, dim3(B), dim3(CB_THREADS), lds, s, pa, pb, dA, iA, dB, iB, gl, Na, Nb, wA,
                       wB, ga);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int R>
static int launch_nn(const float* q, const float* t, int B, int Nq, int Nt, float* d, int32_t* idx, hipStream_t s) {
    int gx = (Nq + 64 * R - 1) / (64 * R);
    VPN_LAUNCH(chamfer_nn_kernel<R>, dim3(gx, B), dim3(CH_BLOCK), 0, s, q, t, Nq, Nt, d, idx);
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


// =====================================================================================
// Pruned exact nearest neighbour.
//
// Same contract as chamfer_nn_kernel (bit-exact d2, IEEE sqrt, lowest ORIGINAL index among
// targets whose sqrt ties), far fewer point pairs:
//   1. cloud_sort_kernel: per sample, counting sort of a cloud by the Morton code of a 16^3 grid over
//      its bounding box -> sorted coordinates, permutation, and a bounding box per chunk of 64
//      consecutive sorted points (a compact blob along the space-filling curve);
//   2. chamfer_nn_pruned_kernel: one wave = 64*R consecutive SORTED queries (a compact set with a
//      box).  Lanes hold the squared box-to-box distance to 64 target chunks; the wave repeatedly
//      takes the nearest remaining chunk (DPP min-reduce + ballot), streams its 64 points through
//      LDS with the same packed inner loop, and stops as soon as the nearest remaining box is
//      farther than the wave's worst current best by more than PRUNE_MARGIN (which covers fp32
//      rounding of the bound and the width of a sqrt bucket, so a skipped target can neither win
//      nor tie).  Ties are resolved on ORIGINAL indices in the epilogue.
// =====================================================================================
constexpr int CP_CELLS = 4096;          // 16 x 16 x 16 Morton cells
constexpr int CP_CHUNK = 64;
constexpr float CP_PRUNE_MARGIN = 2.0e-6f;

#define VPN_DPPF(v, ctrl, rmask) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false))
// wave64 reductions on the VALU (DPP row shifts + row broadcasts); the result is wave-uniform
__device__ inline float wave_min_u(float v) {
    v = fminf(v, VPN_DPPF(v, 0x111, 0xf)); v = fminf(v, VPN_DPPF(v, 0x112, 0xf));
    v = fminf(v, VPN_DPPF(v, 0x114, 0xf)); v = fminf(v, VPN_DPPF(v, 0x118, 0xf));
    v = fminf(v, VPN_DPPF(v, 0x142, 0xa)); v = fminf(v, VPN_DPPF(v, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline float wave_max_u(float v) {
    v = fmaxf(v, VPN_DPPF(v, 0x111, 0xf)); v = fmaxf(v, VPN_DPPF(v, 0x112, 0xf));
    v = fmaxf(v, VPN_DPPF(v, 0x114, 0xf)); v = fmaxf(v, VPN_DPPF(v, 0x118, 0xf));
    v = fmaxf(v, VPN_DPPF(v, 0x142, 0xa)); v = fmaxf(v, VPN_DPPF(v, 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ inline unsigned wave_max_bits(unsigned v) {
#define VPN_DPPU(v, ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, rmask, 0xf, false)
    v = max(v, VPN_DPPU(v, 0x111, 0xf)); v = max(v, VPN_DPPU(v, 0x112, 0xf));
    v = max(v, VPN_DPPU(v, 0x114, 0xf)); v = max(v, VPN_DPPU(v, 0x118, 0xf));
    v = max(v, VPN_DPPU(v, 0x142, 0xa)); v = max(v, VPN_DPPU(v, 0x143, 0xc));
#undef VPN_DPPU
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ inline unsigned spread4(unsigned v) {   // bit i -> bit 3i
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

// one workgroup (1024 lanes) per sample: sorted[b] = points of pts[b] in Morton-cell order,
// perm[b][pos] = original index, boxes[b][c] = (lo xyz, hi xyz) of sorted points [64c, 64c+64)
__global__ __launch_bounds__(1024) void cloud_sort_kernel(const float* __restrict__ pts, int N,
                                                          float* __restrict__ sorted, int32_t* __restrict__ perm,
                                                          float* __restrict__ boxes, int C) {
    __shared__ int hist[CP_CELLS];
    __shared__ float red[6][16];
    __shared__ float bb[6];
    __shared__ int wtot[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)b * N * 3;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = tid; i < N; i += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = p[i * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = wave_min_u(lo[a]), h = wave_max_u(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    for (int i = tid; i < CP_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < 16; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        bb[tid] = v;
    }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) sc[a] = bb[3 + a] > bb[a] ? 15.99f / (bb[3 + a] - bb[a]) : 0.0f;
    auto key_of = [&](float x, float y, float z) -> unsigned {
        const unsigned cx = (unsigned)min(15, max(0, (int)((x - bb[0]) * sc[0])));
        const unsigned cy = (unsigned)min(15, max(0, (int)((y - bb[1]) * sc[1])));
        const unsigned cz = (unsigned)min(15, max(0, (int)((z - bb[2]) * sc[2])));
        return spread4(cx) | (spread4(cy) << 1) | (spread4(cz) << 2);
    };
    for (int i = tid; i < N; i += 1024) atomicAdd(&hist[key_of(p[i * 3], p[i * 3 + 1], p[i * 3 + 2])], 1);
    __syncthreads();
    // exclusive prefix sum over the 4096 cells: 4 cells per lane, wave scan, wave totals
    const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    const int sum4 = h0 + h1 + h2 + h3;
    int inc = sum4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    const int ex = base + inc - sum4;
    hist[4 * tid] = ex; hist[4 * tid + 1] = ex + h0; hist[4 * tid + 2] = ex + h0 + h1; hist[4 * tid + 3] = ex + h0 + h1 + h2;
    __syncthreads();
    float* so = sorted + (size_t)b * N * 3;
    int32_t* po = perm + (size_t)b * N;
    for (int i = tid; i < N; i += 1024) {
        const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
        const int pos = atomicAdd(&hist[key_of(x, y, z)], 1);
        so[pos * 3] = x; so[pos * 3 + 1] = y; so[pos * 3 + 2] = z;
        po[pos] = i;
    }
    __threadfence_block();
    __syncthreads();
    for (int c = wave; c < C; c += 16) {
        const int j = c * CP_CHUNK + lane;
        float l[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float h[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (j < N) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { l[a] = so[j * 3 + a]; h[a] = l[a]; }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lm = wave_min_u(l[a]), hm = wave_max_u(h[a]);
            if (lane == 0) { boxes[((size_t)b * C + c) * 6 + a] = lm; boxes[((size_t)b * C + c) * 6 + 3 + a] = hm; }
        }
    }
}

template <int R>
__global__ __launch_bounds__(64) void chamfer_nn_pruned_kernel(
    const float* __restrict__ sq, const int32_t* __restrict__ permq, const float* __restrict__ st,
    const int32_t* __restrict__ permt, const float* __restrict__ boxes_t, int Nq, int Nt, int Ct,
    float* __restrict__ out_dist, int32_t* __restrict__ out_idx) {
    __shared__ __attribute__((aligned(16))) float tile[3][CP_CHUNK];
    const int b = blockIdx.y, lane = threadIdx.x;
    const float* qb = sq + (size_t)b * Nq * 3;
    const float* tb = st + (size_t)b * Nt * 3;
    const int32_t* pq = permq + (size_t)b * Nq;
    const int32_t* pt = permt + (size_t)b * Nt;
    const float* bx = boxes_t + (size_t)b * Ct * 6;
    const int q0 = blockIdx.x * (64 * R) + lane;

    float ax[R], ay[R], az[R], best[R], second[R];
    int gidx[R];
    float qlo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float qhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int qc = min(q0 + r * 64, Nq - 1);
        ax[r] = qb[qc * 3]; ay[r] = qb[qc * 3 + 1]; az[r] = qb[qc * 3 + 2];
        best[r] = __builtin_inff(); second[r] = __builtin_inff(); gidx[r] = 0;
        qlo[0] = fminf(qlo[0], ax[r]); qlo[1] = fminf(qlo[1], ay[r]); qlo[2] = fminf(qlo[2], az[r]);
        qhi[0] = fmaxf(qhi[0], ax[r]); qhi[1] = fmaxf(qhi[1], ay[r]); qhi[2] = fmaxf(qhi[2], az[r]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { qlo[a] = wave_min_u(qlo[a]); qhi[a] = wave_max_u(qhi[a]); }

    float thr = __builtin_inff();            // worst current best of the wave, inflated by the margin
    for (int w0 = 0; w0 < Ct; w0 += 64) {
        float lb = __builtin_inff();
        if (w0 + lane < Ct) {                // squared box-to-box distance to chunk w0 + lane
            const float* bc = bx + (size_t)(w0 + lane) * 6;
            lb = 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float gap = fmaxf(0.0f, fmaxf(bc[a] - qhi[a], qlo[a] - bc[3 + a]));
                lb += gap * gap;
            }
        }
        while (true) {
            const float m = wave_min_u(lb);
            if (m == __builtin_inff() || !(m <= thr)) break;      // nothing left that can win or tie
            const int sel = __builtin_ctzll(__ballot(lb == m));
            if (lane == sel) lb = __builtin_inff();
            const int cbase = (w0 + sel) * CP_CHUNK;
            // stage the chunk (coalesced), sentinel beyond the end of the cloud
            {
                const int j = cbase + lane;
                float x = 3.0e38f, y = 3.0e38f, z = 3.0e38f;
                if (j < Nt) { x = tb[j * 3]; y = tb[j * 3 + 1]; z = tb[j * 3 + 2]; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                tile[0][lane] = x; tile[1][lane] = y; tile[2][lane] = z;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
#pragma unroll 2
            for (int j = 0; j < CP_CHUNK; j += 8) {
                const float4 Xa = *reinterpret_cast<const float4*>(&tile[0][j]), Xb = *reinterpret_cast<const float4*>(&tile[0][j + 4]);
                const float4 Ya = *reinterpret_cast<const float4*>(&tile[1][j]), Yb = *reinterpret_cast<const float4*>(&tile[1][j + 4]);
                const float4 Za = *reinterpret_cast<const float4*>(&tile[2][j]), Zb = *reinterpret_cast<const float4*>(&tile[2][j + 4]);
                const f2 X[4] = {{Xa.x, Xa.y}, {Xa.z, Xa.w}, {Xb.x, Xb.y}, {Xb.z, Xb.w}};
                const f2 Y[4] = {{Ya.x, Ya.y}, {Ya.z, Ya.w}, {Yb.x, Yb.y}, {Yb.z, Yb.w}};
                const f2 Z[4] = {{Za.x, Za.y}, {Za.z, Za.w}, {Zb.x, Zb.y}, {Zb.z, Zb.w}};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const f2 qx = {ax[r], ax[r]}, qy = {ay[r], ay[r]}, qz = {az[r], az[r]};
                    f2 d[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f2 dx = qx - X[h], dy = qy - Y[h], dz = qz - Z[h];
                        d[h] = ((dx * dx) + (dy * dy)) + (dz * dz);
                    }
                    float m8 = __builtin_fminf(__builtin_fminf(d[0].x, d[0].y), d[1].x);
                    m8 = __builtin_fminf(__builtin_fminf(m8, d[1].y), d[2].x);
                    m8 = __builtin_fminf(__builtin_fminf(m8, d[2].y), d[3].x);
                    m8 = __builtin_fminf(m8, d[3].y);
                    const bool up = m8 < best[r];
                    // second smallest group minimum (counting duplicates): median of (best, second, m8)
                    second[r] = __builtin_amdgcn_fmed3f(best[r], second[r], m8);
                    gidx[r] = up ? (cbase + j) : gidx[r];
                    best[r] = up ? m8 : best[r];
                }
            }
#ifdef VPN_CHAMFER_DEBUG
            if (lane == 0) atomicAdd(&g_dbg[5], 1ull);
#endif
            float bm = best[0];
#pragma unroll
            for (int r = 1; r < R; ++r) bm = fmaxf(bm, best[r]);
            thr = wave_max_u(bm) * (1.0f + CP_PRUNE_MARGIN);
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int qi = q0 + r * 64;
        const float s = sqrtf(best[r]);
        const int g = gidx[r];
        int idx = 0x7fffffff;
        // inside the winning group: lowest original index among the targets whose sqrt equals s
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = min(g + e, Nt - 1);
            const float d2 = dist2_exact(ax[r], ay[r], az[r], tb[j * 3], tb[j * 3 + 1], tb[j * 3 + 2]);
            if (g + e < Nt && sqrtf(d2) == s) idx = min(idx, pt[j]);
        }
        // another group also reaches the minimal sqrt (exact duplicate distances or a sqrt-bucket tie):
        // the wave scans all targets for that query and takes the lowest original index
        unsigned long long need = __ballot(qi < Nq && sqrtf(second[r]) == s);
        while (need) {
            const int src = __builtin_ctzll(need);
            need &= need - 1;
            const float qx = __shfl(ax[r], src, 64), qy = __shfl(ay[r], src, 64), qz = __shfl(az[r], src, 64);
            const float ss = __shfl(s, src, 64);
            int loc = 0x7fffffff;
            for (int base = 0; base < Nt; base += 64) {
                const int j = base + lane;
                if (j < Nt && sqrtf(dist2_exact(qx, qy, qz, tb[j * 3], tb[j * 3 + 1], tb[j * 3 + 2])) == ss) loc = min(loc, pt[j]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) loc = min(loc, __shfl_xor(loc, o, 64));
            if (lane == src) idx = min(idx, loc);
#ifdef VPN_CHAMFER_DEBUG
            if (lane == src) atomicAdd(&g_dbg[4], 1ull);
#endif
        }
        if (qi < Nq) {
            const int o = pq[qi];
            out_dist[(size_t)b * Nq + o] = s;
            out_idx[(size_t)b * Nq + o] = idx;
        }
    }
}

// =====================================================================================
// MFMA-filtered exact nearest neighbour.
//
// The brute-force scan is bound by fp32 VALU issue (8 separately rounded flops per pair are needed for
// bit-exact d2).  Here the matrix pipe does the bulk: with A = (bx, by, bz, |b|^2) per target and
// B = (-2ax, -2ay, -2az, 1) per query, two v_mfma_f32_32x32x2_f32 give t_ij = |b_i|^2 - 2 a_j.b_i for
// 32 targets x 32 queries (= d2 - |a_j|^2 up to a rigorous rounding bound E), on a pipe that runs
// beside the VALU, which only keeps a v_min3 tree per 32x32 block: (smallest t, its block, second
// smallest block minimum).  The exact contract is restored in the epilogue:
//   * the 32 targets of the winning block are re-evaluated with the exact separately-rounded d2 and the
//     sqrt tie rule;
//   * if any other block comes within band = 2E + 4e-6 d2 of the winner, the filter cannot decide and
//     the wave re-scans all targets exactly for that query (same cooperative scan as the tie path).
// Outputs are bit-identical to chamfer_nn_kernel.
// =====================================================================================
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int CM_BLOCK = 256;            // 4 waves x 32 queries
#ifndef CM_WAVES16_N
#define CM_WAVES16_N 8
#endif
// waves per workgroup of the fp16 filter: they share the LDS tiles, so 8 waves halve the tile fetches, LDS writes and
// barriers per query (98.7-99.1 us against 100.1-102.5 with 4 waves, same box, eager launches)
constexpr int CM_WAVES16 = CM_WAVES16_N;
constexpr int CM_BLOCK16 = 64 * CM_WAVES16;
template <int PREC> constexpr int cm_block() { return PREC == 2 ? CM_BLOCK16 : CM_BLOCK; }
constexpr int CM_TILE = 512;             // targets per LDS feature tile of the fp32 filter (16 B per target per buffer)
constexpr float CM_EPS = 8.0f * 5.9604644775390625e-08f;   // 8 * 2^-24: bound on the relative rounding of t_ij (fp32 MFMA chain)
// bf16 variant: 21 exact products accumulated in fp32 (<= 24 * 2^-24), dropped cross terms b2a3+b3a2+b3a3
// (<= 4.2 * 2^-24 |a||b|), rounding of |b|^2 (3 * 2^-24): 32 * 2^-24 covers all of it
constexpr float CM_EPS_BF16 = 32.0f * 5.9604644775390625e-08f;
constexpr int CM_TILE16 = 256;           // targets per LDS tile of bf16 rows
constexpr int CM_ROWW = CM_ROWB / 4;     // the same in 4-byte words
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef short bs4 __attribute__((ext_vector_type(4)));      // operand type of v_mfma_f32_32x32x8_bf16

// ---- fp16 variant of the filter (PREC == 2): fp16 has 11 significant bits against bf16's 8, so TWO pieces per
// coordinate carry 22 bits and the three leading cross terms b1a1, b1a2, b2a1 per coordinate (9) plus three pieces of
// |b|^2 fill 12 of the 16 K slots of ONE v_mfma_f32_32x32x16_f16 per 32x32 block: 32 matrix-pipe cycles instead of
// 48, 32-byte rows instead of 48 (feature kernel writes and tile fetches shrink by a third, one 16-byte operand read
// per lane and block instead of 16 + 8).  fp16 has a narrow exponent range, so the coordinates are scaled by
// S = 2^11 (exact) before the split: |coordinate| <= 8 keeps S x below the fp16 maximum and S^2 |b|^2 within three
// fp16 pieces times fp16-representable powers of two; a cloud or query outside that range (|p|^2 > 64, or not finite)
// makes the query UNDECIDED, i.e. it is resolved exactly by the fix-up kernel: correct for any input, fast for
// normalised shapes.  Error bound CM_EPS_F16 in near_error / the epilogue; DESIGN.md 4.1 has the derivation.
constexpr float CM_INV_S16SQ = 1.0f / (2048.0f * 2048.0f);   // 2^-22: filter values come out scaled by S^2
constexpr float CM_DOMAIN16 = 64.0f;                         // |p|^2 bound of the fp16 filter (|coordinate| <= 8)
// dropped terms b2a2 + rb a + b ra: <= 3 * 2^-22 |A||B| = 24 * 2^-24 |a||b|; accumulation of 12 exact products in fp32:
// <= 12 * 2^-24 (2|a||b| + |b|^2); rounding of |b|^2 itself: 3 * 2^-24 |b|^2  ->  (24/2 + 12) = 24 per 2|a||b|, 15 per |b|^2:
// 26 covers both.  The pieces' absolute floor (fp16 subnormal spacing; the matrix pipe keeps subnormal inputs --
// tools/ubench/mfma_f16_denorm.hip -- but the bound below also covers a flush) adds 2^-24 (|a|_1 + |b|_1).
constexpr float CM_EPS_F16 = 26.0f * 5.9604644775390625e-08f;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// feature planes / rows of a cloud and the kernel body that writes them: vpn_chamfer_feat.h (shared with sampler.hip,
// whose forward launch of the training step writes the features of the cloud it has just sampled)

// One launch converts both clouds of a Chamfer call.  1-D grid of B * (j0.ysplit + j1.ysplit) workgroups, decoded
// such that sample b runs on XCD b / (B/8) — where the filter kernel will read what is written here (its own remap).
__global__ __launch_bounds__(CFEAT_THREADS) void chamfer_feat_kernel(const FeatJob j0, const FeatJob j1, int B) {
    __shared__ float red[CFEAT_THREADS / 64];
    int b, sy;
    feat_decode(blockIdx.x, B, b, sy);
    const bool other = sy >= j0.ysplit;
    feat_slice(other ? j1 : j0, b, sy - (other ? j0.ysplit : 0), red);
}

// Queries the filter could not decide (another 32-target block within the error band of the best one: 0.2-2 %
// of them at C3) are listed per sample by the filter kernel and resolved exactly here.  A workgroup takes CF_Q
// listed queries of ONE sample at a time, so every target it loads (feature planes: SoA, coalesced float4 loads,
// 16 targets per lane in flight) serves CF_Q queries — one query per workgroup re-read the whole target cloud per query, 216 MB
// of L2 traffic per launch at C3.  Per lane and query: minimum d2, its first index, the runner-up value; only if
// some other d2 lies within a few ulp of the minimum (it could share the sqrt) a second pass compares square roots.
constexpr int CF_THREADS = 256;
constexpr int CF_Q = 4;
constexpr int CF_UNROLL = 4;
#ifndef CF_GROUPS_N
#define CF_GROUPS_N 16
#endif
constexpr int CF_GROUPS = CF_GROUPS_N;             // workgroups per sample

// list layout: count[B] then entries[B][Nq] (query numbers of sample b)
struct FixJob {           // one direction of a Chamfer call
    const float* F; int Nq, Nt, Ntp;
    float* out_dist; int32_t* out_idx; const int* undecided;
    const int32_t* perm;  // planes in Morton order: perm[b][pos] = original target index (NULL: planes in original order)
};

// list layout (ints): count[pad4(B)], then entries[B][Nq] of 16 bytes: (query number, x, y, z)
__device__ __host__ inline int pad4(int n) { return (n + 3) & ~3; }

// 1-D grid of 2 * B * CF_GROUPS workgroups.  Workgroups are dealt round-robin to the 8 XCDs; the filter kernel
// runs sample b on XCD b / (B/8) (its own remap), where the sample's list, planes and outputs now sit in L2 —
// read from any other XCD each of the dependent loads below costs ~2 us.  So the linear id is decoded such
// that (sample, group, direction) lands on that XCD again (speed only: any placement is correct).
__global__ __launch_bounds__(CF_THREADS) void chamfer_fixup_kernel(const FixJob j0, const FixJob j1, int B) {
    int b, grp, dir;
    {
        const int id = blockIdx.x, per = B >> 3;
        if ((B & 7) == 0) {
            const int xcd = id & 7, r = id >> 3;
            b = xcd * per + r % per;
            grp = (r / per) % CF_GROUPS;
            dir = r / (per * CF_GROUPS);
        } else {
            b = id % B; grp = (id / B) % CF_GROUPS; dir = id / (B * CF_GROUPS);
        }
    }
    const bool other = dir != 0;
    const float* __restrict__ F = other ? j1.F : j0.F;
    const int Nq = other ? j1.Nq : j0.Nq, Nt = other ? j1.Nt : j0.Nt, Ntp = other ? j1.Ntp : j0.Ntp;
    float* __restrict__ out_dist = other ? j1.out_dist : j0.out_dist;
    int32_t* __restrict__ out_idx = other ? j1.out_idx : j0.out_idx;
    const int* __restrict__ undecided = other ? j1.undecided : j0.undecided;
    const int32_t* __restrict__ perm = other ? j1.perm : j0.perm;
    __shared__ float redf[CF_Q][4];
    __shared__ int redi[CF_Q][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int4* list = reinterpret_cast<const int4*>(undecided + pad4(B)) + (size_t)b * Nq;
    // the count and the first group of entries are read together (entries past the count are stale but in bounds)
    int4 ent[CF_Q];
#pragma unroll
    for (int c = 0; c < CF_Q; ++c) ent[c] = list[min(grp * CF_Q + c, Nq - 1)];
    const int count = undecided[b];
    const float* fx = F + (size_t)b * 4 * Ntp;
    const float* fy = fx + Ntp;
    const float* fz = fy + Ntp;
    const int32_t* pm = perm ? perm + (size_t)b * Ntp : nullptr;
    const bool sorted = pm != nullptr;
    for (int k0 = grp * CF_Q; k0 < count; k0 += CF_GROUPS * CF_Q) {
        int q[CF_Q];
        float qx[CF_Q], qy[CF_Q], qz[CF_Q], best[CF_Q], second[CF_Q];
        int bi[CF_Q];
        if (k0 != grp * CF_Q) {
#pragma unroll
            for (int c = 0; c < CF_Q; ++c) ent[c] = list[min(k0 + c, count - 1)];
        }
        float thr[CF_Q];
#pragma unroll
        for (int c = 0; c < CF_Q; ++c) {
            const int4 e = (k0 + c < count) ? ent[c] : ent[0];        // a short last group repeats a listed query
            q[c] = e.x; qx[c] = __int_as_float(e.y); qy[c] = __int_as_float(e.z); qz[c] = __int_as_float(e.w);
            best[c] = __builtin_inff(); second[c] = __builtin_inff(); bi[c] = 0x7fffffff;
            // The filter kernel left dist = sqrt(m2), m2 = the EXACT minimum over its best block: the true minimum is
            // <= m2, so only targets with d2 <= m2 (1 + 1e-6) can win or tie.  They are found with a contracted (FMA)
            // d2, 6 full-rate instructions per pair, compared once per 4 targets; the exact, separately rounded
            // evaluation with the tie bookkeeping (13 instructions per pair, 5 of them half-rate) runs only for the
            // groups of 4 targets in which some lane of the wave has a candidate (~5 % of them).  The FMA value is
            // within 4e-7 relative of the exact one and dist^2 within 1.2e-7 of m2: 1e-5 covers both with room.
            const float dq = out_dist[(size_t)b * Nq + min(max(q[c], 0), Nq - 1)];
            thr[c] = (dq * dq) * (1.0f + 1.0e-5f);
            thr[c] = thr[c] >= 0.0f ? thr[c] : __builtin_inff();       // NaN distance: examine everything exactly
        }
        // float4 plane loads: 4 consecutive targets per load, CF_UNROLL x 3 loads in flight per lane (the planes are
        // padded to Ntp, a multiple of 64, with sentinel rows that never win; loads are clamped inside them)
        const int n4 = Ntp >> 2;
        for (int base = threadIdx.x; base < n4; base += CF_UNROLL * CF_THREADS) {
            float4 x[CF_UNROLL], y[CF_UNROLL], z[CF_UNROLL];
            int4 o[CF_UNROLL];
#pragma unroll
            for (int u = 0; u < CF_UNROLL; ++u) {
                const int g4 = min(base + u * CF_THREADS, n4 - 1);
                x[u] = reinterpret_cast<const float4*>(fx)[g4];
                y[u] = reinterpret_cast<const float4*>(fy)[g4];
                z[u] = reinterpret_cast<const float4*>(fz)[g4];
                o[u] = pm ? reinterpret_cast<const int4*>(pm)[g4] : make_int4(g4 * 4, g4 * 4 + 1, g4 * 4 + 2, g4 * 4 + 3);
            }
#pragma unroll
            for (int u = 0; u < CF_UNROLL; ++u) {
                const int j0 = (base + u * CF_THREADS) * 4;
                const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
                const float zs[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
                const int os[4] = {o[u].x, o[u].y, o[u].z, o[u].w};
                bool cand = false;
#pragma unroll
                for (int c = 0; c < CF_Q; ++c) {
                    float dt[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dx = qx[c] - xs[e], dy = qy[c] - ys[e], dz = qz[c] - zs[e];
                        dt[e] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    }
                    cand |= fminf(fminf(fminf(dt[0], dt[1]), dt[2]), dt[3]) <= thr[c];
                }
                if (!__builtin_amdgcn_ballot_w64(cand)) continue;      // wave-uniform: nobody has a candidate in these 4 targets
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // rows past Nt (padding, clamped loads) are pushed to +inf by an additive penalty: written as a
                    // conditional, every evaluation became an exec-mask branch
                    const float pen = (j0 + e < Nt) ? 0.0f : __builtin_inff();
#pragma unroll
                    for (int c = 0; c < CF_Q; ++c) {
                        const float d = dist2_exact(qx[c], qy[c], qz[c], xs[e], ys[e], zs[e]) + pen;
                        second[c] = __builtin_amdgcn_fmed3f(best[c], second[c], d);   // a duplicate minimum counts as runner-up
                        // lowest ORIGINAL index among equal minima (in original order that is simply the first one)
                        const bool take = (d < best[c]) | (sorted & (d == best[c]) & (os[e] < bi[c]));
                        bi[c] = take ? os[e] : bi[c];
                        best[c] = fminf(best[c], d);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CF_Q; ++c) {
            const float mw = wave_min_u(best[c]);
            if (lane == 0) redf[c][wave] = mw;
        }
        __syncthreads();
        float s[CF_Q], lim[CF_Q];
        int loc[CF_Q];
        bool close = false;
#pragma unroll
        for (int c = 0; c < CF_Q; ++c) {
            const float m = fminf(fminf(redf[c][0], redf[c][1]), fminf(redf[c][2], redf[c][3]));
            s[c] = sqrtf(m); lim[c] = m * (1.0f + 1.0e-6f);
            // another lane's minimum within a few ulp of m can share its sqrt: that lane's candidate is known (bi),
            // no rescan needed.  Only two candidates inside ONE lane (runner-up within the limit too) need the rescan.
            loc[c] = (best[c] == m || (best[c] <= lim[c] && sqrtf(best[c]) == s[c])) ? bi[c] : 0x7fffffff;
            close |= second[c] <= lim[c];
        }
        if (__syncthreads_or(close)) {
            for (int j = threadIdx.x; j < Nt; j += CF_THREADS) {
                const float x = fx[j], y = fy[j], z = fz[j];
                const int oj = pm ? pm[j] : j;
#pragma unroll
                for (int c = 0; c < CF_Q; ++c) {
                    const float d = dist2_exact(qx[c], qy[c], qz[c], x, y, z);
                    if (d <= lim[c] && sqrtf(d) == s[c]) loc[c] = min(loc[c], oj);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CF_Q; ++c) {
            int l = loc[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) l = min(l, __shfl_xor(l, o, 64));
            if (lane == 0) redi[c][wave] = l;
        }
        __syncthreads();
        if (threadIdx.x < CF_Q && k0 + (int)threadIdx.x < count) {
            const int c = threadIdx.x;
            float sc = s[0]; int qc = q[0];
#pragma unroll
            for (int e = 1; e < CF_Q; ++e) { sc = c == e ? s[e] : sc; qc = c == e ? q[e] : qc; }
            // the query number comes from a list entry: entries below `count` were written by the scan kernel, but a
            // store address must not depend on that alone (an ablation build that dropped the `count` guard above
            // faulted on a stale entry: DESIGN.md Finding 4)
            qc = min(max(qc, 0), Nq - 1);
            const int found = min(min(redi[c][0], redi[c][1]), min(redi[c][2], redi[c][3]));
            if (found != 0x7fffffff) {                 // nothing found only for NaN / infinite inputs: the filter's answer stays
                out_dist[(size_t)b * Nq + qc] = sc;
                out_idx[(size_t)b * Nq + qc] = found;
            }
        }
        __syncthreads();                                   // redf / redi are rewritten by the next group
    }
}

// balanced v_min3 tree (depth 3): a serial chain of 8 dependent v_min3 made the filter latency-bound.
// Level 1 reads the matrix-pipe output and stays visible to the compiler (it owns the MFMA -> VALU wait
// states; inline asm is not covered by its hazard recogniser): five fminf pairs that it fuses into v_min3.
// The upper levels are written as instructions: left to the compiler, the tree is re-associated so that one
// 2-operand v_min_f32 reads raw accumulators and needs two extra NaN-quieting v_max per call.
__device__ inline float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ inline float min16(const f16v& v) {
    const float a = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]);
    const float b = __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]);
    const float c = __builtin_fminf(__builtin_fminf(v[6], v[7]), v[8]);
    const float d = __builtin_fminf(__builtin_fminf(v[9], v[10]), v[11]);
    const float e = __builtin_fminf(__builtin_fminf(v[12], v[13]), v[14]);
    // v[15] is read by an asm instruction, but one that also consumes d and e: it cannot issue before they have
    const float g = vmin3(d, e, v[15]);
    return vmin3(vmin3(a, b, c), g, g);
}

// the same over the 32 accumulators of two blocks: 16 instructions (10 compiler-visible first-level v_min3, then 6)
__device__ inline float min32(const f16v& u, const f16v& v) {
    const float a = __builtin_fminf(__builtin_fminf(u[0], u[1]), u[2]);
    const float b = __builtin_fminf(__builtin_fminf(u[3], u[4]), u[5]);
    const float c = __builtin_fminf(__builtin_fminf(u[6], u[7]), u[8]);
    const float d = __builtin_fminf(__builtin_fminf(u[9], u[10]), u[11]);
    const float e = __builtin_fminf(__builtin_fminf(u[12], u[13]), u[14]);
    const float a2 = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]);
    const float b2 = __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]);
    const float c2 = __builtin_fminf(__builtin_fminf(v[6], v[7]), v[8]);
    const float d2 = __builtin_fminf(__builtin_fminf(v[9], v[10]), v[11]);
    const float e2 = __builtin_fminf(__builtin_fminf(v[12], v[13]), v[14]);
    const float g = vmin3(d, e, u[15]), g2 = vmin3(d2, e2, v[15]);
    const float h = vmin3(a, b, c), h2 = vmin3(a2, b2, c2);
    return vmin3(vmin3(h, g, h2), g2, g2);
}

// Filter error of the targets that can still WIN OR TIE once the exact minimum m2 of the best block is known: such a
// target b has exact d2(a, b) <= m2 (1 + 1e-6), hence |b| <= |a| + sqrt(m2) (triangle inequality): its error bound
// needs the query's norm and m2 only, not the largest norm of the cloud -- for a query near the centre of the cloud
// the band shrinks ~2x and with it the number of undecided queries.  (Rigour: DESIGN.md 4.1; the factors 1 + 2e-6
// cover the fp32 rounding of the two square roots and of the sum.)
// sna, sm2: UPPER bounds of sqrt(na), sqrt(m2) good to ~1e-6 relative (sqrt_up below, or IEEE sqrt): a correctly
// rounded sqrt is a ~12-instruction sequence here and this runs once per query in an issue-bound kernel.
__device__ inline float near_error(float eps, float abs_eps, float na, float sna, float sm2, float E_cloud) {
    const float rb = (sna + sm2) * (1.0f + 2.0e-6f);
    const float nbn = rb * rb;
    // sqrt(na nbn) = sqrt(na) rb <= sna rb;  abs_eps (|a|_1 + |b|_1) <= abs_eps sqrt(3) (|a| + |b|): the fp16 pieces'
    // absolute floor (0 for the other filters)
    const float En = (eps * (2.0f * sna * rb + nbn + na) + abs_eps * 1.7320509f * (sna + rb)) * (1.0f + 2.0e-6f);
    return fminf(En, E_cloud);
}
// upper bound of sqrt(x), x >= 0: v_sqrt_f32 is good to 1 ulp (6e-8); the factor covers it and the roundings of the
// products it enters, the constant a flushed denormal input (sqrt of a denormal < 1.1e-19)
__device__ inline float sqrt_up(float x) { return __builtin_amdgcn_sqrtf(x) * (1.0f + 4.0e-7f) + 1.0e-18f; }


// Exact resolution of a scan workgroup's own undecided queries (0.4 % of the queries at C3: 3.6 per workgroup of
// direction 2, 0.2 of direction 1) by the workgroup itself, THREADS lanes, before it exits -- this used to be a launch
// of its own (17-20 us at C3) that re-scanned the clouds from lists in memory.  FQ listed queries at a time against the
// whole target cloud (fp32 planes x, y, z: coalesced float4 loads, UN groups of 4 targets per lane in flight):
//   prefilter: t = |b|^2 - 2 a.b as three FMAs per pair.  t = d2 - |a|^2 within E = CM_EPS (|a| + |b|)^2 (3 roundings in
//     |b|^2, 3 in the chain), and a target that can win or tie has exact d2 <= m2 (1 + 1e-6) (m2 = the exact
//     minimum over the filter's best cell >= the true minimum), hence t <= thr = (m2 (1 + 1e-5) - |a|^2) + E + 4e-7 (m2 + |a|^2);
//   a candidate (a handful per query) is evaluated exactly -- separately rounded d2, IEEE sqrt -- and posts the key
//     (sqrt bits, index) to an LDS 64-bit atomic min: the smallest key IS the reference's answer, the lowest index among
//     the targets whose SQUARE ROOT is smallest (chamfer_distance.py:19-23).  No per-lane bookkeeping, no second pass.
// The list lives in the workgroup's LDS: lqd = (query x, y, z, the filter's sqrt(m2)), lqi = (query number, the
// filter's index); snb = an upper bound of the largest target norm.
#ifndef CM_FQ
#define CM_FQ 4
#endif
#ifndef CM_FUN
#define CM_FUN 2
#endif
template <int THREADS>
__device__ __forceinline__ void fixup_own(const float* __restrict__ Fb, float snb, int Nq, int Nt, int Ntp,
                                 float* __restrict__ out_dist, int32_t* __restrict__ out_idx,
                                 float4* lqd, const int2* lqi, int count) {
    constexpr int FQ = THREADS >= 512 ? CM_FQ : 2;   // listed queries per pass over the targets (register budget of the host kernel)
    constexpr int UN = CM_FUN;           // groups of 4 targets in flight per lane: a pass is latency, not arithmetic
    __shared__ unsigned long long key[FQ];
    const int n4 = Ntp >> 2;
    const float4* f4 = reinterpret_cast<const float4*>(Fb);          // planes x, y, z at f4 + {0, 1, 2} * n4
    if (threadIdx.x < FQ) key[threadIdx.x] = ~0ull;
    for (int k0 = 0; k0 < count; k0 += FQ) {
        // the queries of a pass are the same for every lane: m = -2 a (the query is -0.5 m, exactly) and the threshold
        // live in SGPRs, which leaves the vector registers to the target loads in flight
        float mx[FQ], my[FQ], mz[FQ], thr[FQ];
#pragma unroll
        for (int c = 0; c < FQ; ++c) {
            const int k = min(k0 + c, count - 1);
            const float4 a = lqd[k];                                  // the query and sqrt(m2)
            const float na = a.x * a.x + a.y * a.y + a.z * a.z;
            const float sab = sqrt_up(na) + snb;
            const float E = CM_EPS * (sab * sab) * (1.0f + 1.0e-6f);
            const float m2 = a.w * a.w;                               // within 1.2e-7 relative of the exact m2
            float th = (m2 * (1.0f + 1.0e-5f) - na) + (E + 4.0e-7f * (m2 + na));
            th = th == th ? th : __builtin_inff();                    // NaN anywhere: every target is examined exactly
            th = k0 + c < count ? th : -__builtin_inff();             // slots past the end of the list: no candidates
            auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
            mx[c] = uni(-2.0f * a.x); my[c] = uni(-2.0f * a.y); mz[c] = uni(-2.0f * a.z); thr[c] = uni(th);
        }
        __syncthreads();                                              // keys initialised
        for (int base = threadIdx.x; base < n4; base += UN * THREADS) {
            float4 x[UN], y[UN], z[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {                            // rows past the cloud: clamped loads, skipped below
                const int g4 = min(base + u * THREADS, n4 - 1);
                x[u] = f4[g4]; y[u] = f4[n4 + g4]; z[u] = f4[2 * n4 + g4];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (base + u * THREADS >= n4) break;
                const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
                const float zs[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
                float ws[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)                           // |b|^2: 3 roundings
                    ws[e] = __builtin_fmaf(zs[e], zs[e], __builtin_fmaf(ys[e], ys[e], xs[e] * xs[e]));
#pragma unroll
                for (int c = 0; c < FQ; ++c) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        t[e] = __builtin_fmaf(mx[c], xs[e], __builtin_fmaf(my[c], ys[e], __builtin_fmaf(mz[c], zs[e], ws[e])));
                    if (fminf(fminf(fminf(t[0], t[1]), t[2]), t[3]) <= thr[c]) {      // rare: a handful of lanes per query
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = (base + u * THREADS) * 4 + e;
                            if (t[e] <= thr[c] && j < Nt) {                            // rows past Nt are padding
                                const float d = dist2_exact(-0.5f * mx[c], -0.5f * my[c], -0.5f * mz[c], xs[e], ys[e], zs[e]);
                                atomicMin(&key[c], ((unsigned long long)__float_as_uint(sqrtf(d)) << 32) | (unsigned)j);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < FQ) {
            const int k = k0 + threadIdx.x;
            const unsigned long long best = key[threadIdx.x];
            key[threadIdx.x] = ~0ull;                                // for the next group (ordered by its barrier)
            if (k < count) {
                const int qc = min(max(lqi[k].x, 0), Nq - 1);
                // no candidate at all only for NaN / infinite inputs: the filter's answer stands
                const float df = best != ~0ull ? __uint_as_float((unsigned)(best >> 32)) : lqd[k].w;
                out_dist[qc] = df;
                out_idx[qc] = best != ~0ull ? (int)(unsigned)best : lqi[k].y;
                lqd[k].w = df;                                       // the caller sums the final distances
            }
        }
    }
}

struct ScanJob {          // one direction of a Chamfer call
    const float* qpts; const float* F; const unsigned short* H; const unsigned int* nmax;
    int Nq, Nt, Ntp, gx, G;                       // G = gx * B workgroups
    float* out_dist; int32_t* out_idx;
    float* wgsum;                                 // [B][gx]: sum of the minima of each workgroup's queries (fixed order)
};

// Both directions of a Chamfer call in ONE launch: workgroups [0, j0.G) run job 0, the rest job 1.  Launched
// back to back each direction had its own tail (at C3 4096 workgroups on 1280 resident slots = 3.2 rounds, then
// 1024 = 0.8 rounds: 5 rounds of time); together they pack into 4.  The long job (more targets per query) goes
// first so that its workgroups start early and the short ones fill in behind.
template <int PREC>   // 0: fp32-input MFMA (shares the fp32 vector datapath), 1: bf16 3-piece split on the matrix pipe
// 6 waves per SIMD: the tile loop needs ~70 VGPRs and LDS admits 3 workgroups of 8 waves per CU; the bound keeps the
// fix-up tail (which few workgroups run) from raising the whole kernel's register allocation
#ifndef CM_WAVES_PER_EU
#define CM_WAVES_PER_EU 6
#endif
__global__ __launch_bounds__(cm_block<PREC>(), PREC == 1 ? 5 : CM_WAVES_PER_EU) void chamfer_nn_mfma_kernel(
    const ScanJob j0, const ScanJob j1, int nsamples, const RasterOrderJob oj) {
    // LDS of the fp16 filter's tiles, declared here because the rider below borrows it
    __shared__ __attribute__((aligned(16))) unsigned char tileH16[PREC == 2 ? 2 * CM_TILE16 * CM_ROWB : 16];
    if constexpr (PREC == 2) {
        // Rider of the training step: the workgroups BEHIND the two scan jobs (the last to be dispatched, i.e. in the tail
        // of the launch, where a third of the slots is idle) sort the raster's tiles, one image each.  The launch sits
        // between the one that writes the raster records and the one that reads masks and order, so neither a hand-off
        // nor a launch of its own is needed.
        // the rider (~10 us since it also tests the quadrants) is dispatched FIRST: ids [0, oj.B).  Last, it was the tail of
        // the launch (+10 us); first it costs 64 of 768 slots for its lifetime (not measurable)
        if ((int)blockIdx.x < oj.B) {
            __builtin_amdgcn_s_setprio(3);               // few waves with a long dependent chain among scan waves at full tilt
            raster_order_wg<cm_block<PREC>()>(oj, (int)blockIdx.x, tileH16);
            return;
        }
    }
    const int wg = (int)blockIdx.x - (PREC == 2 ? oj.B : 0);
#ifdef CM_EXP_TRACE
    const unsigned long long t_start = wall_clock64();
#endif
    __shared__ int s_cnt;                                           // length of the workgroup's list of undecided queries (epilogue)
    if (threadIdx.x == 0) s_cnt = 0;
    const bool other = wg >= j0.G;
    const float* __restrict__ qpts = other ? j1.qpts : j0.qpts;
    const float* __restrict__ F = other ? j1.F : j0.F;
    const unsigned short* __restrict__ H = other ? j1.H : j0.H;
    const unsigned int* __restrict__ nmax = other ? j1.nmax : j0.nmax;
    const int Nq = other ? j1.Nq : j0.Nq, Nt = other ? j1.Nt : j0.Nt, Ntp = other ? j1.Ntp : j0.Ntp;
    const int gx = other ? j1.gx : j0.gx, G = other ? j1.G : j0.G;
    float* __restrict__ out_dist = other ? j1.out_dist : j0.out_dist;
    int32_t* __restrict__ out_idx = other ? j1.out_idx : j0.out_idx;
    float* __restrict__ wgsum = other ? j1.wgsum : j0.wgsum;
    // Workgroups are dealt round-robin over the 8 XCDs (L2 is per XCD), so the id inside the job is remapped such
    // that all workgroups of a sample land on ONE XCD and stream its target rows out of that XCD's L2 (speed only:
    // any placement is correct; j0.G is a multiple of 8 whenever B is, so id & 7 is still the XCD).
    int vid = wg - (other ? j0.G : 0);
    {
        const int per = G >> 3;
        if (vid < (per << 3)) vid = (vid & 7) * per + (vid >> 3);
    }
    const int b = vid / gx, bx = vid - b * gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jq = lane & 31, half = lane >> 5;
    const float* qb = qpts + (size_t)b * Nq * 3;
    const int qi = (bx * (cm_block<PREC>() / 64) + wave) * 32 + jq;
    const int qc = min(qi, Nq - 1);
    const F3 a3 = ld3(qb + qc * 3);
    const float ax = a3.x, ay = a3.y, az = a3.z;
    // B operand (K x N): lane supplies B[k = lane/32][n = lane%32]
    const float bq0 = half ? -2.0f * ay : -2.0f * ax;
    const float bq1 = half ? 1.0f : -2.0f * az;
    // A operand (M x K): lane supplies A[i = lane%32][k = lane/32] from the feature planes (through LDS)

    // smallest and second smallest block minimum of this lane's half and the block of the smallest.  (Tracking the
    // second block and a third value too cost 8 VALU instructions per block instead of 4, in a loop that is bound
    // by VALU issue; queries whose runner-up block is within the error band now go to the fix-up list instead.)
    float best = __builtin_inff(), second = __builtin_inff();
    int blk = 0;
// branch-free selects (the compiler otherwise turns the index updates into exec-mask branches)
#define CM_SEL(dst, cond_mask, a, b) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dst) : "v"(b), "v"(a), "s"(cond_mask))
#define CM_UPDATE(m, tb0)                                                                      \
    {                                                                                          \
        const int t_ = (tb0);                                                                  \
        unsigned long long k1_;                                                                \
        asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(k1_) : "v"(m), "v"(best));               \
        second = __builtin_amdgcn_fmed3f(best, second, (m));                                   \
        CM_SEL(blk, k1_, t_, blk);        /* c1 ? t : blk */                                   \
        asm volatile("v_min_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(m));                 \
    }
// the same with the block number inside the tile as an inline constant (no v_mov of the index per block)
#define CM_UPDATE_C(m, J)                                                                      \
    {                                                                                          \
        unsigned long long k1_;                                                                \
        asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(k1_) : "v"(m), "v"(best));               \
        second = __builtin_amdgcn_fmed3f(best, second, (m));                                   \
        asm volatile("v_cndmask_b32 %0, %1, " #J ", %2" : "=v"(blkc) : "v"(blkc), "s"(k1_));   \
        asm volatile("v_min_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(m));                 \
    }
    const float* Fb = F + (size_t)b * 4 * Ntp;
    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 2) {
        // ---- fp16 filter: ONE v_mfma_f32_32x32x16_f16 per 32x32 block (12 of 16 K slots used), rows of 32 bytes
        unsigned char (*tileH)[CM_TILE16 * CM_ROWB] = reinterpret_cast<unsigned char (*)[CM_TILE16 * CM_ROWB]>(tileH16);   // [2][...], LDS stride 48 B per row
        const unsigned char* Hb = reinterpret_cast<const unsigned char*>(H) + (size_t)b * Ntp * CM_ROWB16;
        h8 bq;                                       // this lane's query: K slots [8 half, +8)
        {
            _Float16 qx[2], qy[2], qz[2];
            split2_f16(-2.0f * CM_S16 * ax, qx); split2_f16(-2.0f * CM_S16 * ay, qy); split2_f16(-2.0f * CM_S16 * az, qz);
            // K slot k of the query row: per coordinate the pieces (A1 A2 A1) against the target's (b1 b1 b2), then the
            // multipliers of the three pieces of S^2 |b|^2, then zeros.  Compile-time indices: the row lives in registers.
            auto slot = [&](int k) -> _Float16 {
                if (k < 9) {
                    const int c = k / 3, t = k % 3;
                    const _Float16 v0 = c == 0 ? qx[0] : (c == 1 ? qy[0] : qz[0]);
                    const _Float16 v1 = c == 0 ? qx[1] : (c == 1 ? qy[1] : qz[1]);
                    return t == 1 ? v1 : v0;
                }
                return k == 9 ? (_Float16)32768.0f : (k == 10 ? (_Float16)16.0f : (k == 11 ? (_Float16)1.0f : (_Float16)0.0f));
            };
#pragma unroll
            for (int e = 0; e < 8; ++e) bq[e] = half ? slot(8 + e) : slot(e);
        }
        constexpr int F4 = CM_TILE16 * CM_ROWB16 / 16 / CM_BLOCK16;    // float4 per lane per tile
        static_assert(F4 == 2 || F4 == 1, "tile fetch is written for 1 or 2 float4 per lane");
        // tile t0 = rows [t0, t0 + CM_TILE16): one contiguous 8-KB run in HBM; float4 f of it is half (f & 1) of row f >> 1
        // and goes to LDS byte (f >> 1) * 48 + (f & 1) * 16.  No bounds checks: the row buffer is sized for 48-byte rows.
        const float4* Hq = reinterpret_cast<const float4*>(Hb) + threadIdx.x;
        struct Pre { float4 a, b; };
        auto fetch = [&](int t0) -> Pre {
            const float4* p = Hq + t0 * (CM_ROWB16 / 16);
            if constexpr (F4 == 2) return Pre{p[0], p[CM_BLOCK16]};
            else return Pre{p[0], p[0]};
        };
        const int f0 = threadIdx.x, f1 = threadIdx.x + CM_BLOCK16;
        const int o0 = (f0 >> 1) * CM_ROWB + (f0 & 1) * 16, o1 = (f1 >> 1) * CM_ROWB + (f1 & 1) * 16;
        auto stash = [&](int buf, const Pre& v) {
            *reinterpret_cast<float4*>(&tileH[buf][o0]) = v.a;
            if constexpr (F4 == 2) *reinterpret_cast<float4*>(&tileH[buf][o1]) = v.b;
        };
        auto rd = [&](const unsigned char* T, int blkk) -> float4 {
            return *reinterpret_cast<const float4*>(T + blkk * 32 * CM_ROWB);
        };
        auto block = [&](const float4& o) {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, o), bq, zero, 0, 0, 0);
        };
        static_assert(CM_TILE16 == 256, "the tile loop below is unrolled for 8 blocks per tile");
// The tracked unit is a PAIR of blocks (64 targets): one 16-instruction tree over both accumulators and one
// 4-instruction update per pair, 10 VALU instructions per block instead of 12 in a loop bound by VALU issue; the
// exact finish then works on the winning half-wave's 32 rows of the pair (CM_CELL below).
#define CM_PRIO(p)
#define CM_PAIR(oa, ob, J)                                                                     \
    {                                                                                          \
        CM_PRIO(1)                                                                             \
        const f16v accA = block(oa), accB = block(ob);                                         \
        CM_PRIO(0)                                                                             \
        const float mAB = min32(accA, accB);                                                   \
        CM_UPDATE_C(mAB, J)                                                                    \
    }
        Pre pre = fetch(0);
        stash(0, pre);
        __syncthreads();
        int buf = 0, blkc = 0;
        for (int t0 = 0; t0 < Ntp; t0 += CM_TILE16, buf ^= 1) {
            const bool more = t0 + CM_TILE16 < Ntp;
            if (more) pre = fetch(t0 + CM_TILE16);     // in flight during the MFMA loop, stored to LDS after it
            const int nblk = min(CM_TILE16, Ntp - t0) >> 5;          // 2, 4, 6 or 8 (Ntp is a multiple of 64)
            const unsigned char* T = &tileH[buf][jq * CM_ROWB + half * 16];
            const float before = best;
            float4 a0 = rd(T, 0), a1 = rd(T, 1), b0 = rd(T, 2), b1 = rd(T, 3);
            CM_PAIR(a0, a1, 0)
            if (nblk > 2) {
                a0 = rd(T, 4); a1 = rd(T, 5);
                CM_PAIR(b0, b1, 1)
                if (nblk > 4) {
                    b0 = rd(T, 6); b1 = rd(T, 7);
                    CM_PAIR(a0, a1, 2)
                    if (nblk > 6) CM_PAIR(b0, b1, 3)
                }
            }
            blk = best < before ? t0 + (blkc << 6) : blk;            // the tile improved this lane's minimum
            if (more) stash(buf ^ 1, pre);
            __syncthreads();
        }
#undef CM_PAIR
    } else     if constexpr (PREC == 1) {
        // ---- bf16 filter: v_mfma_f32_32x32x16_bf16 (K slots 0..15) + v_mfma_f32_32x32x8_bf16 (K slots 16..23) per 32x32
        //      block, on the matrix pipe beside the VALU
        __shared__ __attribute__((aligned(16))) unsigned char tileH[2][CM_TILE16 * CM_ROWB];
        const unsigned char* Hb = reinterpret_cast<const unsigned char*>(H) + (size_t)b * Ntp * CM_ROWB;
        bf8 bqA;                                     // this lane's query: K slots [8 half, +8) of the first MFMA
        bs4 bqB;                                     //                    K slots 16 + [4 half, +4) of the second
        {
            unsigned short qx[3], qy[3], qz[3];
            split3_bf16(-2.0f * ax, qx); split3_bf16(-2.0f * ay, qy); split3_bf16(-2.0f * az, qz);
            // K slot k of the query row: per coordinate the pieces (a1 a2 a1 a3 a1 a2), then 1.0 x 3, then zeros.
            // Everything is indexed by compile-time constants so the row lives in registers (no scratch).
            auto slot = [&](int k) -> unsigned short {
                const int c = k / 6, t = k % 6;
                const int pi = (t == 1 || t == 5) ? 1 : (t == 3 ? 2 : 0);
                if (k < 18) {
                    const unsigned short v0 = c == 0 ? qx[0] : (c == 1 ? qy[0] : qz[0]);
                    const unsigned short v1 = c == 0 ? qx[1] : (c == 1 ? qy[1] : qz[1]);
                    const unsigned short v2 = c == 0 ? qx[2] : (c == 1 ? qy[2] : qz[2]);
                    return pi == 0 ? v0 : (pi == 1 ? v1 : v2);
                }
                return k < 21 ? (unsigned short)0x3F80 : (unsigned short)0;      // bf16 1.0 for the |b|^2 pieces
            };
            us8 ua;
            us4 ub;
#pragma unroll
            for (int e = 0; e < 8; ++e) ua[e] = half ? slot(8 + e) : slot(e);
#pragma unroll
            for (int e = 0; e < 4; ++e) ub[e] = half ? slot(20 + e) : slot(16 + e);
            bqA = __builtin_bit_cast(bf8, ua); bqB = __builtin_bit_cast(bs4, ub);
        }
        constexpr int F4 = CM_TILE16 * CM_ROWB / 16 / CM_BLOCK;        // float4 per lane per tile
        static_assert(F4 == 3, "tile fetch is written for 3 float4 per lane");
        // tile t0 = rows [t0, t0 + CM_TILE16): one contiguous 12-KB run, copied as it is (the LDS stride is the row
        // size); lane i moves float4 i, i+256, i+512.  No bounds checks (they cost ~50 instructions per tile): the last
        // tile may run past Ntp into rows that are allocated (next sample / slack at the end of the workspace),
        // loaded and never used.
        const float4* Hq = reinterpret_cast<const float4*>(Hb) + threadIdx.x;
        struct Pre { float4 a, b, c; };                 // named members: an indexed array ended up in scratch
        auto fetch = [&](int t0) -> Pre {
            const float4* p = Hq + t0 * (CM_ROWB / 16);
            return Pre{p[0], p[CM_BLOCK], p[2 * CM_BLOCK]};
        };
        auto stash = [&](int buf, const Pre& v) {
            float4* q = reinterpret_cast<float4*>(&tileH[buf][0]) + threadIdx.x;
            q[0] = v.a; q[CM_BLOCK] = v.b; q[2 * CM_BLOCK] = v.c;
        };
        // one 32-target block: lane (jq, half) reads bytes [16 half, +16) and [32 + 8 half, +8) of row jq
        struct Ops { float4 lo; float2 hi; };
        auto rd = [&](const unsigned char* T, int blkk) -> Ops {
            const unsigned char* p = T + blkk * 32 * CM_ROWB;
            return Ops{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float2*>(p + 32 - 8 * half)};
        };
        auto block = [&](const Ops& o) {
            f16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, o.lo), bqA, zero, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(bs4, o.hi), bqB, acc, 0, 0, 0);
            return acc;
        };
        static_assert(CM_TILE16 == 256, "the tile loop below is unrolled for 8 blocks per tile");
#define CM_PAIR(oa, ob, JA, JB)                                                                \
    {                                                                                          \
        const f16v accA = block(oa), accB = block(ob);                                         \
        const float mA = min16(accA), mB = min16(accB);                                        \
        CM_UPDATE_C(mA, JA)                                                                    \
        CM_UPDATE_C(mB, JB)                                                                    \
    }
        Pre pre = fetch(0);
        stash(0, pre);
        __syncthreads();
        int buf = 0, blkc = 0;
        for (int t0 = 0; t0 < Ntp; t0 += CM_TILE16, buf ^= 1) {
            const bool more = t0 + CM_TILE16 < Ntp;
            if (more) pre = fetch(t0 + CM_TILE16);     // in flight during the MFMA loop, stored to LDS after it
            const int nblk = min(CM_TILE16, Ntp - t0) >> 5;          // 2, 4, 6 or 8 (Ntp is a multiple of 64)
            const unsigned char* T = &tileH[buf][jq * CM_ROWB + half * 16];
            const float before = best;
            // two operand register sets in ping-pong; rows past nblk are in the tile, read and not used
            Ops a0 = rd(T, 0), a1 = rd(T, 1), b0 = rd(T, 2), b1 = rd(T, 3);
            CM_PAIR(a0, a1, 0, 1)
            if (nblk > 2) {
                a0 = rd(T, 4); a1 = rd(T, 5);
                CM_PAIR(b0, b1, 2, 3)
                if (nblk > 4) {
                    b0 = rd(T, 6); b1 = rd(T, 7);
                    CM_PAIR(a0, a1, 4, 5)
                    if (nblk > 6) CM_PAIR(b0, b1, 6, 7)
                }
            }
            blk = best < before ? t0 + (blkc << 5) : blk;            // the tile improved this lane's minimum
            if (more) stash(buf ^ 1, pre);
            __syncthreads();
        }
#undef CM_PAIR
    } else {
    // The 4 waves of the workgroup scan the same targets: feature tiles of CM_TILE targets go through LDS
    // (double buffered: the next tile's loads are in flight while this one feeds the matrix pipe).
    __shared__ __attribute__((aligned(16))) float tileF[2][4][CM_TILE];
    // each lane moves 2 float4 per tile (4 planes x CM_TILE floats = 512 float4, 256 lanes)
    constexpr int CM_F4 = CM_TILE / CM_BLOCK;       // float4 per lane per tile
    auto fetch = [&](int t0, float4 v[CM_F4]) {
#pragma unroll
        for (int u = 0; u < CM_F4; ++u) {
            const int i = threadIdx.x + u * CM_BLOCK;
            const int plane = i / (CM_TILE / 4), off = (i - plane * (CM_TILE / 4)) * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 3.0e38f);
            if (t0 + off < Ntp) v[u] = *reinterpret_cast<const float4*>(Fb + (size_t)plane * Ntp + t0 + off);
        }
    };
    auto stash = [&](int buf, const float4 v[CM_F4]) {
#pragma unroll
        for (int u = 0; u < CM_F4; ++u) {
            const int i = threadIdx.x + u * CM_BLOCK;
            const int plane = i / (CM_TILE / 4), off = (i - plane * (CM_TILE / 4)) * 4;
            *reinterpret_cast<float4*>(&tileF[buf][plane][off]) = v[u];
        }
    };
    float4 pre[CM_F4];
    fetch(0, pre);
    stash(0, pre);
    __syncthreads();
    int buf = 0;
    for (int t0 = 0; t0 < Ntp; t0 += CM_TILE, buf ^= 1) {
        const bool more = t0 + CM_TILE < Ntp;
        if (more) fetch(t0 + CM_TILE, pre);         // in flight during the MFMA loop, stored to LDS after it
        const int nblk = min(CM_TILE, Ntp - t0) >> 5;
        const float* T0 = &tileF[buf][half][jq];
        const float* T1 = &tileF[buf][2 + half][jq];
        // operands of the next pair of blocks are read from LDS while the current MFMAs run
        float o0 = T0[0], o1 = T1[0], o2 = T0[32], o3 = T1[32];
        for (int u = 0; u < nblk; u += 2) {         // nblk is even (tiles are multiples of 64 targets)
            const float c0 = o0, c1 = o1, c2 = o2, c3 = o3;
            if (u + 2 < nblk) { o0 = T0[u * 32 + 64]; o1 = T1[u * 32 + 64]; o2 = T0[u * 32 + 96]; o3 = T1[u * 32 + 96]; }
            f16v accA = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, bq0, zero, 0, 0, 0);   // two independent accumulators:
            f16v accB = __builtin_amdgcn_mfma_f32_32x32x2f32(c2, bq0, zero, 0, 0, 0);   // the min-tree of one block
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, bq1, accA, 0, 0, 0);         // overlaps the MFMAs of the other
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(c3, bq1, accB, 0, 0, 0);
            const float mA = min16(accA);           // this lane: query jq, 16 of the block's 32 targets
            CM_UPDATE(mA, t0 + u * 32)
            const float mB = min16(accB);
            CM_UPDATE(mB, t0 + u * 32 + 32)
        }
        if (more) stash(buf ^ 1, pre);
        __syncthreads();                            // everybody done with `buf`, next tile landed
    }
    }
#ifdef CM_EXP_TRACE
    const unsigned long long t_loop = wall_clock64();
#endif
    // The tracked unit (cell) is what ONE half-wave saw of CM_CELL consecutive targets: lane (jq, half) holds the
    // accumulator rows 8 g + 4 half + r (g = 0 .. CM_CELL/8 - 1, r = 0..3) of every block.  The winner is the cell with
    // the smallest filtered minimum over both half-waves of the query; V2 = the smallest filtered minimum over ALL other
    // cells (the loser's best, the winner's second).  Only the winner cell is evaluated exactly, its CM_CELL/2 targets
    // split over the two lanes of the query: half of the exact work of evaluating the whole block (or pair of blocks).
    constexpr int CM_CELL = PREC == 2 ? 64 : 32;       // targets per tracked unit: the fp16 loop tracks pairs of blocks
    constexpr int NT = CM_CELL / 4;                    // exact evaluations per lane
    int K, hw;
    float Bv, V2;
    {
        const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
        const int ok = __shfl_xor(blk, 32, 64);
        // lexicographic (value, block, half): both lanes of the query reach the same verdict
        const bool mine = best < ob || (best == ob && (blk < ok || (blk == ok && half == 0)));
        Bv = mine ? best : ob;
        K = mine ? blk : ok;
        hw = mine ? half : half ^ 1;
        V2 = mine ? fminf(second, ob) : fminf(os, best);
    }
    float d2[NT];
    float m2 = __builtin_inff();
#ifdef VPN_CHAMFER_DEBUG
    if (K < 0 || K + CM_CELL > Ntp || (K & (CM_CELL - 1))) atomicAdd(&g_dbg[7], 1ull);    // the cell index the gather below trusts
#endif
    // K is a multiple of CM_CELL below Ntp by construction (tile base + unit number inside the tile; Ntp is a multiple
    // of 64); the clamp costs one instruction per query and keeps the gather inside the padded planes whatever
    // happened upstream
    K = min(max(K, 0), Ntp - CM_CELL);
    // this lane's targets: groups g = half * NT/4 + j (j < NT/4) of 4 consecutive targets at K + 8 g + 4 hw
    const int base = K + half * (2 * NT) + 4 * hw;
    {
        float4 X[NT / 4], Y[NT / 4], Z[NT / 4];
#pragma unroll
        for (int v = 0; v < NT / 4; ++v) {
            X[v] = *reinterpret_cast<const float4*>(Fb + base + 8 * v);
            Y[v] = *reinterpret_cast<const float4*>(Fb + (size_t)Ntp + base + 8 * v);
            Z[v] = *reinterpret_cast<const float4*>(Fb + 2 * (size_t)Ntp + base + 8 * v);
        }
#pragma unroll
        for (int v = 0; v < NT / 4; ++v) {
            const float xs[4] = {X[v].x, X[v].y, X[v].z, X[v].w}, ys[4] = {Y[v].x, Y[v].y, Y[v].z, Y[v].w};
            const float zs[4] = {Z[v].x, Z[v].y, Z[v].z, Z[v].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) d2[v * 4 + w] = dist2_exact(ax, ay, az, xs[w], ys[w], zs[w]);
        }
        // padding targets (coordinates 0) are masked only by the waves whose winner cell reaches into the padding:
        // three instructions per target that the other waves (all of them when Nt is a multiple of 64) skip
        if (__ballot(K + CM_CELL > Nt)) {
#pragma unroll
            for (int e = 0; e < NT; ++e) d2[e] += (base + 8 * (e >> 2) + (e & 3) < Nt) ? 0.0f : __builtin_inff();
        }
#pragma unroll
        for (int e = 0; e < NT; ++e) m2 = fminf(m2, d2[e]);
    }
    m2 = fminf(m2, __shfl_xor(m2, 32, 64));
    // t = d2 - |a|^2 within E; exact d2 within 4e-7 relative: a block whose filtered minimum lies outside `band`
    // can neither win nor tie.  Otherwise the query is undecided here and goes to the fix-up list.
    static_assert(CFEAT_SLOTS <= 64, "one slot per lane");
    // norms are >= 0: their bit patterns order like unsigned integers (v_max_u32 takes the DPP operand directly and
    // needs no NaN quieting; a NaN norm has the largest pattern and so survives, as it must)
    const float nb = __uint_as_float(wave_max_bits(lane < CFEAT_SLOTS ? nmax[b * CFEAT_SLOTS + lane] : 0u));
    const float na = ax * ax + ay * ay + az * az;
    const float sna = sqrt_up(na), snb = sqrt_up(nb);                           // bounds only: no IEEE sqrt sequences
    float E = (PREC == 2 ? CM_EPS_F16 : (PREC == 1 ? CM_EPS_BF16 : CM_EPS)) * (2.0f * sna * snb + nb + na) * (1.0f + 1.0e-6f);
    if (PREC == 2) {
        E += 5.9604644775390625e-08f * 1.7320509f * (sna + snb);               // absolute floor of the fp16 pieces
        Bv *= CM_INV_S16SQ; V2 *= CM_INV_S16SQ;                                 // the fp16 filter works on S x: values scaled by S^2
    }
    // m2 is exact, so the runner-up is compared with it rather than with the filtered value of the best block:
    // only the runner-up's own filter error E remains (the band was 2E before; this halves the undecided queries).
    // 4e-6 m2 covers the rounding of m2 and of |a|^2 and the width of a sqrt bucket.
    const float s = sqrtf(m2);                                      // IEEE: this one is the result
    const float band = E + 4.0e-6f * (m2 + na);                     // any target of the cloud (consistency of the best block)
    // possible winners: a runner-up that can win or tie has exact d2 <= m2 (1 + 1e-6), i.e. a real d2 <= m2 (1 + 1.4e-6),
    // and filters to at most that - |a|^2 + E_near; against the computed m2 - na (|a|^2 good to 3 ulp, the difference to
    // 1): 4e-6 m2 + 4e-7 na covers it (|a|^2 used to be charged 4e-6 too: 2e-6 of a 4e-6 band for a query half a unit from
    // the origin -- and 40 % of the undecided queries of direction 2)
    const float band_near = near_error(PREC == 2 ? CM_EPS_F16 : (PREC == 1 ? CM_EPS_BF16 : CM_EPS), PREC == 2 ? 5.9604644775390625e-08f : 0.0f,
                                       na, sna, s * (1.0f + 1.0e-7f), E) + (4.0e-6f * m2 + 4.0e-7f * na);
    bool ambiguous = !(V2 > (m2 - na) + band_near) || !(m2 - na <= Bv + band);
    if (PREC == 2) ambiguous |= !(nb <= CM_DOMAIN16 && na <= CM_DOMAIN16);     // outside the fp16 filter's range: exact fix-up
    // lowest index attaining the minimum; a different d2 can only share the sqrt if it lies within a few ulp
    // of it, which is rare: only then are the sqrt values compared
    const float lim = m2 * (1.0f + 1.0e-6f);
    int li = NT;                                                    // position inside this lane's targets (inline constants)
#pragma unroll
    for (int e = NT - 1; e >= 0; --e)
        if (d2[e] == m2) li = e;                                    // target index grows with e: the lowest wins
    int idx = li < NT ? base + 8 * (li >> 2) + (li & 3) : 0x7fffffff;
    // near tie: some d2 with m2 < d2 <= lim.  d2 >= 0, so the bit patterns order like the values: the smallest
    // (bits(d2) - bits(m2) - 1) as an unsigned number is below bits(lim) - bits(m2) exactly then (equal values wrap
    // to 0xffffffff; a NaN d2 has a larger pattern than any finite lim).  One subtraction per target and a min3 tree
    // instead of three compares and two mask operations per target.
    bool near_tie;
    {
        const unsigned mb1 = __float_as_uint(m2) + 1u;
        unsigned gm = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < NT; e += 2)                              // the compiler forms v_min3_u32
            gm = min(gm, min(__float_as_uint(d2[e]) - mb1, __float_as_uint(d2[e + 1]) - mb1));
        near_tie = m2 < __builtin_inff() && gm < __float_as_uint(lim) - __float_as_uint(m2);
    }
    if (__ballot(near_tie)) {
#pragma unroll
        for (int e = NT - 1; e >= 0; --e)
            if (d2[e] <= lim && sqrtf(d2[e]) == s) idx = min(idx, base + 8 * (e >> 2) + (e & 3));
    }
    idx = min(idx, __shfl_xor(idx, 32, 64));
    // Decided queries store their result.  An undecided one goes on the WORKGROUP's list in LDS as (query, sqrt(m2),
    // index inside the best cell) and is resolved exactly by this workgroup before it exits (fixup_own): no list in
    // memory, no second launch, nothing shared between workgroups.
    constexpr int QPW = cm_block<PREC>() / 2;                       // queries per workgroup = capacity of the list
    __shared__ float4 s_qd[QPW];
    __shared__ int2 s_qi[QPW];
    __shared__ float s_ws[cm_block<PREC>() / 64];
    int pos = -1;                                                   // (s_cnt was zeroed before the tile loop, whose barriers order it)
    if (half == 0 && qi < Nq) {
        if (!ambiguous) {
            out_dist[(size_t)b * Nq + qi] = s;
            out_idx[(size_t)b * Nq + qi] = idx;
        } else {
            pos = atomicAdd(&s_cnt, 1);
            s_qd[pos] = make_float4(ax, ay, az, s); s_qi[pos] = make_int2(qi, idx);
#ifdef VPN_CHAMFER_DEBUG
            atomicAdd(&g_dbg[6], 1ull);
#endif
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
#ifdef CM_EXP_TRACE
    const unsigned long long t_epi = wall_clock64();
#endif
    float sfin = (half == 0 && qi < Nq) ? s : 0.0f;                 // this query's final minimum (once per query)
    if (cnt != 0) {
#ifdef VPN_CHAMFER_DEBUG
        const unsigned long long t_fix = wall_clock64();
#endif
        fixup_own<cm_block<PREC>()>(Fb, snb, Nq, Nt, Ntp, out_dist + (size_t)b * Nq, out_idx + (size_t)b * Nq, s_qd, s_qi, cnt);
        __syncthreads();
        if (pos >= 0) sfin = s_qd[pos].w;                           // the resolved distance of an undecided query
#ifdef VPN_CHAMFER_DEBUG
        if (threadIdx.x == 0) {   // [1]/[2]: 100 MHz ticks spent in the fix-up by workgroups of job 0 / job 1, [4]/[5]: how many
            atomicAdd(&g_dbg[other ? 2 : 1], wall_clock64() - t_fix);
            atomicAdd(&g_dbg[other ? 5 : 4], 1ull);
        }
#endif
    }
    // sum of the workgroup's minima in a fixed order (lanes by butterfly, waves in order): the loss finalisation adds
    // gx numbers per sample and direction instead of re-reading dist [B,Nq]
    sfin = wave_sum(sfin);
    if (lane == 0) s_ws[wave] = sfin;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = s_ws[0];
#pragma unroll
        for (int v = 1; v < cm_block<PREC>() / 64; ++v) t += s_ws[v];
        wgsum[(size_t)b * gx + bx] = t;
#ifdef CM_EXP_TRACE
        if (PREC == 2 && wg < 8192) {
            unsigned long long* g = g_ctrace + (size_t)wg * 8;
            g[0] = t_start; g[1] = t_loop; g[2] = t_epi; g[3] = wall_clock64(); g[4] = other ? 1ull : 0ull; g[5] = (unsigned long long)cnt;
        }
#endif
    }
}

// =====================================================================================
// Sorted + pruned variant of the matrix-pipe filter (mode 5).
//
// Both clouds are put in Morton order once (chamfer_sortfeat_kernel: feature planes, bf16 rows, permutation and a
// bounding box per 32-target block, all in sorted order).  A wave then owns 32 CONSECUTIVE SORTED queries — a
// compact set with a box — and runs the same two-MFMA block filter only over the target blocks whose box can
// still hold a target as near as the wave's worst current best: first the nearest block, then, 64 blocks at a
// time, every block with  box-to-box distance^2 <= max_q(best_q) (+ the filter error and a rounding margin).
// A skipped block can neither win nor tie (same argument as CP_PRUNE_MARGIN above).  Each wave streams its
// blocks straight from L2 with a 4-deep prefetch ring; there are no LDS tiles and no workgroup barriers.
// The exact finish and the undecided list are those of chamfer_nn_mfma_kernel, on ORIGINAL indices.
// =====================================================================================
constexpr int CS_THREADS = 1024;
constexpr int CS_BOXF = 8;                         // floats per block box: lo xyz, hi xyz, 2 pad

struct SortJob {
    const float* pts; int N, Np;
    float* F; unsigned int* nmax; unsigned short* H; int* undecided; int32_t* perm; float* boxes;
};

// 1-D grid of 2 * B workgroups (both clouds of a Chamfer call), sample b on XCD b / (B/8) like the other kernels.
__global__ __launch_bounds__(CS_THREADS) void chamfer_sortfeat_kernel(const SortJob j0, const SortJob j1, int B) {
    __shared__ int hist[CP_CELLS];
    __shared__ float red[7][CS_THREADS / 64];
    __shared__ float bb[7];
    __shared__ int wtot[CS_THREADS / 64];
    int b, which;
    {
        const int id = blockIdx.x, per = B >> 3;
        if ((B & 7) == 0) { const int xcd = id & 7, r = id >> 3; b = xcd * per + r % per; which = r / per; }
        else { b = id % B; which = id / B; }
    }
    const bool other = which != 0;
    const float* __restrict__ pts = other ? j1.pts : j0.pts;
    const int N = other ? j1.N : j0.N, Np = other ? j1.Np : j0.Np;
    float* __restrict__ F = other ? j1.F : j0.F;
    unsigned int* __restrict__ nmax = other ? j1.nmax : j0.nmax;
    unsigned short* __restrict__ H = other ? j1.H : j0.H;
    int* __restrict__ undecided = other ? j1.undecided : j0.undecided;
    int32_t* __restrict__ perm = other ? j1.perm : j0.perm;
    float* __restrict__ boxes = other ? j1.boxes : j0.boxes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)b * N * 3;
    if (tid == 0) undecided[b] = 0;
    // ---- bounding box and max norm
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float nv = 0.0f;
    for (int i = tid; i < N; i += CS_THREADS) {
        const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
        lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
        nv = fmaxf(nv, x * x + y * y + z * z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // wave_min_u / wave_max_u order floats as given (DPP fmin/fmax): fine for signed values too
        const float l = wave_min_u(lo[a]), h = wave_max_u(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    {
        const float m = wave_max_u(nv);
        if (lane == 0) red[6][wave] = m;
    }
    for (int i = tid; i < CP_CELLS; i += CS_THREADS) hist[i] = 0;
    __syncthreads();
    if (tid < 7) {
        float v = red[tid][0];
        for (int w = 1; w < CS_THREADS / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        bb[tid] = v;
    }
    __syncthreads();
    if (tid < CFEAT_SLOTS) nmax[b * CFEAT_SLOTS + tid] = tid == 0 ? __float_as_uint(bb[6]) : 0u;
    float sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) sc[a] = bb[3 + a] > bb[a] ? 15.99f / (bb[3 + a] - bb[a]) : 0.0f;
    auto key_of = [&](float x, float y, float z) -> unsigned {
        const unsigned cx = (unsigned)min(15, max(0, (int)((x - bb[0]) * sc[0])));
        const unsigned cy = (unsigned)min(15, max(0, (int)((y - bb[1]) * sc[1])));
        const unsigned cz = (unsigned)min(15, max(0, (int)((z - bb[2]) * sc[2])));
        return spread4(cx) | (spread4(cy) << 1) | (spread4(cz) << 2);
    };
    // ---- counting sort by Morton cell (the order inside a cell is whatever the LDS atomics give: every
    //      grouping is exact, the results do not depend on it)
    for (int i = tid; i < N; i += CS_THREADS) atomicAdd(&hist[key_of(p[i * 3], p[i * 3 + 1], p[i * 3 + 2])], 1);
    __syncthreads();
    {
        static_assert(CP_CELLS == 4 * CS_THREADS, "4 cells per lane in the prefix sum");
        const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const int sum4 = h0 + h1 + h2 + h3;
        int inc = sum4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        const int ex = base + inc - sum4;
        hist[4 * tid] = ex; hist[4 * tid + 1] = ex + h0; hist[4 * tid + 2] = ex + h0 + h1; hist[4 * tid + 3] = ex + h0 + h1 + h2;
    }
    __syncthreads();
    float* f = F + (size_t)b * 4 * Np;
    int32_t* po = perm + (size_t)b * Np;
    for (int i = tid; i < Np; i += CS_THREADS) {
        float x = 0.f, y = 0.f, z = 0.f, n = 3.0e38f;      // padding: a sentinel that never wins, kept at the end
        int pos = i;
        if (i < N) {
            x = p[i * 3]; y = p[i * 3 + 1]; z = p[i * 3 + 2];
            n = x * x + y * y + z * z;
            pos = atomicAdd(&hist[key_of(x, y, z)], 1);
        }
        f[pos] = x; f[Np + pos] = y; f[2 * Np + pos] = z; f[3 * Np + pos] = n;
        po[pos] = i < N ? i : 0x7fffffff;
        write_row(H, (size_t)b * Np + pos, x, y, z, n);
    }
    __threadfence_block();
    __syncthreads();
    // ---- a box per block of 32 consecutive sorted points (padding excluded; an all-padding block gets an empty
    //      box, +inf / -inf, whose distance to anything is +inf)
    const int nb32 = Np >> 5;
    for (int c = wave; c < nb32; c += CS_THREADS / 64) {
        const int j = c * 32 + (lane & 31);
        const bool ok = j < N;
        const float x = ok ? f[j] : 0.f, y = ok ? f[Np + j] : 0.f, z = ok ? f[2 * Np + j] : 0.f;
        const float inf = __builtin_inff();
        const float lx = wave_min_u(ok ? x : inf), ly = wave_min_u(ok ? y : inf), lz = wave_min_u(ok ? z : inf);
        const float hx = wave_max_u(ok ? x : -inf), hy = wave_max_u(ok ? y : -inf), hz = wave_max_u(ok ? z : -inf);
        if (lane == 0) {
            float4* o = reinterpret_cast<float4*>(boxes + ((size_t)b * nb32 + c) * CS_BOXF);
            o[0] = make_float4(lx, ly, lz, hx);
            o[1] = make_float4(hy, hz, 0.f, 0.f);
        }
    }
}

constexpr int CPR_RING = 4;                        // blocks in flight per wave

__global__ __launch_bounds__(CM_BLOCK) void chamfer_nn_mfma_pruned_kernel(
    const float* __restrict__ Fq, const int32_t* __restrict__ permq, int Nqp,      // queries: sorted planes + permutation
    const float* __restrict__ F, const unsigned short* __restrict__ H, const unsigned int* __restrict__ nmax,
    const int32_t* __restrict__ permt, const float* __restrict__ boxes,           // targets, sorted
    int Nq, int Nt, int Ntp, int gx, float* __restrict__ out_dist, int32_t* __restrict__ out_idx,
    int* __restrict__ undecided, int nsamples) {
    int vid = blockIdx.x;
    {
        const int G = gridDim.x, per = G >> 3;
        if (vid < (per << 3)) vid = (vid & 7) * per + (vid >> 3);
    }
    const int b = vid / gx, bx = vid - b * gx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jq = lane & 31, half = lane >> 5;
    const int pos = (bx * 4 + wave) * 32 + jq;                     // sorted position of this lane's query
    if ((bx * 4 + wave) * 32 >= Nq) return;                        // whole wave past the end (no barriers in here)
    const int pc = min(pos, Nq - 1);
    const float* fq = Fq + (size_t)b * 4 * Nqp;
    const float ax = fq[pc], ay = fq[Nqp + pc], az = fq[2 * Nqp + pc];
    float best = __builtin_inff(), second = __builtin_inff();
    int blk = 0;
    const float* Fb = F + (size_t)b * 4 * Ntp;
    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf8 bqA;
    bs4 bqB;
    {
        unsigned short qx[3], qy[3], qz[3];
        split3_bf16(-2.0f * ax, qx); split3_bf16(-2.0f * ay, qy); split3_bf16(-2.0f * az, qz);
        auto slot = [&](int k) -> unsigned short {
            const int c = k / 6, t = k % 6;
            const int pi = (t == 1 || t == 5) ? 1 : (t == 3 ? 2 : 0);
            if (k < 18) {
                const unsigned short v0 = c == 0 ? qx[0] : (c == 1 ? qy[0] : qz[0]);
                const unsigned short v1 = c == 0 ? qx[1] : (c == 1 ? qy[1] : qz[1]);
                const unsigned short v2 = c == 0 ? qx[2] : (c == 1 ? qy[2] : qz[2]);
                return pi == 0 ? v0 : (pi == 1 ? v1 : v2);
            }
            return k < 21 ? (unsigned short)0x3F80 : (unsigned short)0;
        };
        us8 ua;
        us4 ub;
#pragma unroll
        for (int e = 0; e < 8; ++e) ua[e] = half ? slot(8 + e) : slot(e);
#pragma unroll
        for (int e = 0; e < 4; ++e) ub[e] = half ? slot(20 + e) : slot(16 + e);
        bqA = __builtin_bit_cast(bf8, ua); bqB = __builtin_bit_cast(bs4, ub);
    }
    float nb = 0.0f;
#pragma unroll
    for (int w = 0; w < CFEAT_SLOTS; ++w) nb = fmaxf(nb, __uint_as_float(nmax[b * CFEAT_SLOTS + w]));
    const float na = ax * ax + ay * ay + az * az;
    const float E = CM_EPS_BF16 * (2.0f * sqrtf(na * nb) + nb + na);
    // box of the wave's queries (lanes past the end repeat the last query)
    const float qlx = wave_min_u(ax), qly = wave_min_u(ay), qlz = wave_min_u(az);
    const float qhx = wave_max_u(ax), qhy = wave_max_u(ay), qhz = wave_max_u(az);
    const int nblk = Ntp >> 5;
    const float* bxs = boxes + (size_t)b * nblk * CS_BOXF;
    auto box_d2 = [&](int k) -> float {            // squared distance between the query box and the box of block k
        const float4 u = *reinterpret_cast<const float4*>(bxs + (size_t)k * CS_BOXF);
        const float4 v = *reinterpret_cast<const float4*>(bxs + (size_t)k * CS_BOXF + 4);
        const float dx = fmaxf(0.0f, fmaxf(u.x - qhx, qlx - u.w));
        const float dy = fmaxf(0.0f, fmaxf(u.y - qhy, qly - v.x));
        const float dz = fmaxf(0.0f, fmaxf(u.z - qhz, qlz - v.y));
        return dx * dx + dy * dy + dz * dz;
    };
    const unsigned char* Hb = reinterpret_cast<const unsigned char*>(H) + ((size_t)b * Ntp + jq) * CM_ROWB;
    struct Ops { float4 lo; float2 hi; };
    auto ld = [&](int k) -> Ops {
        const unsigned char* r = Hb + (size_t)k * 32 * CM_ROWB;
        return Ops{*reinterpret_cast<const float4*>(r + 16 * half), *reinterpret_cast<const float2*>(r + 32 + 8 * half)};
    };
    auto process = [&](const Ops& o, int k) {
        f16v acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, o.lo), bqA, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(bs4, o.hi), bqB, acc, 0, 0, 0);
        const float m = min16(acc);
        CM_UPDATE(m, k << 5)
    };
    // ---- the nearest block first
    int home;
    {
        float hb = __builtin_inff();
        int hk = 0;
        for (int k = lane; k < nblk; k += 64) {
            const float d = box_d2(k);
            if (d < hb) { hb = d; hk = k; }
        }
        const float m = wave_min_u(hb);
        const unsigned long long at = __ballot(hb == m);
        home = __builtin_amdgcn_readlane(hk, (int)__builtin_ctzll(at));
        process(ld(home), home);
    }
    // ---- then, 64 blocks at a time, every block that can still matter
    for (int base = 0; base < nblk; base += 64) {
        const int k = base + lane;
        const float d = k < nblk ? box_d2(k) : __builtin_inff();
        const float bq = fminf(best, __shfl_xor(best, 32, 64));               // this query's best over both halves
        const float T = wave_max_u(bq + na + E);                              // >= the final d2 of every query here
        unsigned long long m = __ballot(k < nblk && k != home && d <= T * (1.0f + 4.0e-6f));
        const int cnt = __builtin_popcountll(m);
#ifdef VPN_CHAMFER_DEBUG
        if (lane == 0) { atomicAdd(&g_dbg[0], (unsigned long long)cnt); atomicAdd(&g_dbg[1], (unsigned long long)min(64, nblk - base)); }
#endif
        Ops ring[CPR_RING];
        int kk[CPR_RING];
#pragma unroll
        for (int r = 0; r < CPR_RING; ++r) {
            kk[r] = 0;
            if (r < cnt) { kk[r] = base + (int)__builtin_ctzll(m); m &= m - 1; ring[r] = ld(kk[r]); }
        }
        for (int i = 0; i < cnt; i += CPR_RING) {
#pragma unroll
            for (int r = 0; r < CPR_RING; ++r) {
                if (i + r < cnt) {
                    process(ring[r], kk[r]);
                    if (i + r + CPR_RING < cnt) { kk[r] = base + (int)__builtin_ctzll(m); m &= m - 1; ring[r] = ld(kk[r]); }
                }
            }
        }
    }
    // ---- merge the half-waves, exact finish of the best block (as in chamfer_nn_mfma_kernel), original indices
    int K;
    float Bv, V2;
    {
        const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
        const int ok = __shfl_xor(blk, 32, 64);
        Bv = fminf(best, ob);
        K = (ob < best || (ob == best && ok < blk)) ? ok : blk;
        V2 = fminf(K == blk ? second : best, K == ok ? os : ob);
    }
    float d2[16];
    int og[16];
    float m2 = __builtin_inff();
    {
        const int base = K + half * 16;
        const float4* px4 = reinterpret_cast<const float4*>(Fb + base);
        const float4* py4 = reinterpret_cast<const float4*>(Fb + (size_t)Ntp + base);
        const float4* pz4 = reinterpret_cast<const float4*>(Fb + 2 * (size_t)Ntp + base);
        const int4* pm4 = reinterpret_cast<const int4*>(permt + (size_t)b * Ntp + base);
        float4 X[4], Y[4], Z[4];
        int4 P[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { X[v] = px4[v]; Y[v] = py4[v]; Z[v] = pz4[v]; P[v] = pm4[v]; }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float xs[4] = {X[v].x, X[v].y, X[v].z, X[v].w}, ys[4] = {Y[v].x, Y[v].y, Y[v].z, Y[v].w};
            const float zs[4] = {Z[v].x, Z[v].y, Z[v].z, Z[v].w};
            const int ps[4] = {P[v].x, P[v].y, P[v].z, P[v].w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int e = v * 4 + w;
                og[e] = ps[w];                                          // 0x7fffffff on padding rows
                d2[e] = dist2_exact(ax, ay, az, xs[w], ys[w], zs[w]) + ((base + e < Nt) ? 0.0f : __builtin_inff());   // no branch
                m2 = fminf(m2, d2[e]);
            }
        }
    }
    m2 = fminf(m2, __shfl_xor(m2, 32, 64));
    const float band = E + 4.0e-6f * (m2 + na);
    const float s = sqrtf(m2);
    const float band_near = near_error(CM_EPS_BF16, 0.0f, na, sqrt_up(na), s * (1.0f + 1.0e-7f), E) + 4.0e-6f * (m2 + na);
    const bool ambiguous = !(V2 > (m2 - na) + band_near) || !(m2 - na <= Bv + band);
    const float lim = m2 * (1.0f + 1.0e-6f);
    int idx = 0x7fffffff;
    bool near_tie = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (d2[e] == m2) idx = min(idx, og[e]);
        near_tie |= (d2[e] != m2) && (d2[e] <= lim);
    }
    if (__ballot(near_tie)) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (d2[e] <= lim && sqrtf(d2[e]) == s) idx = min(idx, og[e]);
    }
    idx = min(idx, __shfl_xor(idx, 32, 64));
    if (half == 0 && pos < Nq) {
        const int oq = permq[(size_t)b * Nqp + pos];                    // original query number
        out_dist[(size_t)b * Nq + oq] = s;
        out_idx[(size_t)b * Nq + oq] = idx;
        if (ambiguous)
            reinterpret_cast<int4*>(undecided + pad4(nsamples))[(size_t)b * Nq + atomicAdd(undecided + b, 1)] =
                make_int4(oq, __float_as_int(ax), __float_as_int(ay), __float_as_int(az));
    }
}

static inline int pad32(int n) { return (n + 63) & ~63; }   // feature planes padded to 64 targets (blocks are processed in pairs)
// one direction: fp32 planes + bf16 rows of the targets + one tile of slack (the row tiles are fetched without
// bounds checks) + nmax[B][CFEAT_SLOTS] + the lists of undecided queries (count[pad4(B)], 16-byte entries[B][Nq])
static inline int wgsum_per_sample(int Nq) { return (Nq + 127) / 128; }       // capacity: the 4-wave filters have 128 queries per workgroup
static inline size_t mfma_ws_floats(int B, int Nt, int Nq) {
    return (size_t)B * (4 + CM_ROWW) * pad32(Nt) + (size_t)CM_TILE16 * CM_ROWW + (size_t)B * CFEAT_SLOTS + (size_t)pad4(B) + 4 * (size_t)B * Nq
           + (size_t)B * pad32(Nt) + (size_t)B * (pad32(Nt) / 32) * CS_BOXF    // + permutation + block boxes (sorted mode); multiple of 4 floats
           + (size_t)pad4(B * wgsum_per_sample(Nq));                           // + per-workgroup sums of the minima
}

static inline int feat_split(int Ntp) { const int y = (Ntp + CFEAT_WGPTS - 1) / CFEAT_WGPTS; return y > CFEAT_SLOTS ? CFEAT_SLOTS : y; }
struct MfmaWs { float* F; unsigned short* H; unsigned int* nmax; int* undecided; int32_t* perm; float* boxes; float* wgsum; int Ntp; };

static MfmaWs mfma_carve(float* F, int B, int Nt, int Nq) {
    MfmaWs w;
    w.Ntp = pad32(Nt);
    w.F = F;
    w.H = reinterpret_cast<unsigned short*>(F + (size_t)B * 4 * w.Ntp);                       // [B][Ntp][24] bf16
    w.nmax = reinterpret_cast<unsigned int*>(F + (size_t)B * (4 + CM_ROWW) * w.Ntp + (size_t)CM_TILE16 * CM_ROWW);
    w.undecided = reinterpret_cast<int*>(w.nmax + (size_t)B * CFEAT_SLOTS);
    w.perm = reinterpret_cast<int32_t*>(w.undecided + pad4(B) + 4 * (size_t)B * Nq);
    w.boxes = reinterpret_cast<float*>(w.perm + (size_t)B * w.Ntp);
    w.wgsum = w.boxes + (size_t)B * (w.Ntp / 32) * CS_BOXF;      // [B][gx] sums of the minima of the scan against this cloud
    return w;
}

// both directions: features of both clouds (one launch) -> filtered scan of p1 against p2 and of p2 against p1 with
// the exact fix-up of the undecided queries inside (one launch)
static int mfma_both(const float* p1, const float* p2, int B, int N, int M, float* ws, float* d1, int32_t* i1, float* d2,
                     int32_t* i2, int prec, hipStream_t s, bool features_ready = false,
                     const RasterOrderJob* rider = nullptr) {   // prec 0: fp32 MFMA, 1: bf16 x 3, 2: fp16 x 2 (the only one with a rider)
    const bool fp32_filter = prec == 0;
    const MfmaWs w2 = mfma_carve(ws, B, M, N);                            // p2 = targets of direction 1
    const MfmaWs w1 = mfma_carve(ws + mfma_ws_floats(B, M, N), B, N, M);  // p1 = targets of direction 2
    auto split = [](int Ntp) { return feat_split(Ntp); };
    const FeatJob f2{p2, M, w2.Ntp, split(w2.Ntp), w2.F, w2.nmax, fp32_filter ? nullptr : w2.H, prec == 2};
    const FeatJob f1{p1, N, w1.Ntp, split(w1.Ntp), w1.F, w1.nmax, fp32_filter ? nullptr : w1.H, prec == 2};
    hipError_t e = hipSuccess;
    if (!features_ready) {      // mode 7: written by vpn_hotpath_sample_fwd into the same workspace, earlier on this stream
        VPN_LAUNCH(chamfer_feat_kernel, dim3(B * (f2.ysplit + f1.ysplit)), dim3(CFEAT_THREADS), 0, s, f2, f1, B);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    {
        const int qpw = prec == 2 ? 32 * CM_WAVES16 : 128;          // queries per workgroup
        const int gx1 = (N + qpw - 1) / qpw, gx2 = (M + qpw - 1) / qpw;
        const ScanJob s1{p1, w2.F, w2.H, w2.nmax, N, M, w2.Ntp, gx1, gx1 * B, d1, i1, w2.wgsum};     // p1 against p2
        const ScanJob s2{p2, w1.F, w1.H, w1.nmax, M, N, w1.Ntp, gx2, gx2 * B, d2, i2, w1.wgsum};     // p2 against p1
        const bool long_first = (long long)N > (long long)M;       // direction 2 scans the N targets: more work per workgroup
        const ScanJob& ja = long_first ? s2 : s1;
        const ScanJob& jb = long_first ? s1 : s2;
        const RasterOrderJob none{};
        if (prec == 0)
            VPN_LAUNCH(chamfer_nn_mfma_kernel<0>, dim3(ja.G + jb.G), dim3(CM_BLOCK), 0, s, ja, jb, B, none);
        else if (prec == 1)
            VPN_LAUNCH(chamfer_nn_mfma_kernel<1>, dim3(ja.G + jb.G), dim3(CM_BLOCK), 0, s, ja, jb, B, none);
        else
            VPN_LAUNCH(chamfer_nn_mfma_kernel<2>, dim3(ja.G + jb.G + (rider ? rider->B : 0)), dim3(CM_BLOCK16), 0, s, ja, jb, B,
                       rider ? *rider : none);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    // no fix-up launch: every scan workgroup resolves its own undecided queries exactly before it exits (fixup_own)
    return 0;
}

// the two feature jobs of the fp16 filter as mfma_both lays them out (vpn_chamfer_feat.h)
static int chamfer_mode();
int chamfer_feat_jobs(void* workspace, size_t workspace_bytes, int B, int N, int M, const float* p1, const float* p2,
                      FeatJob* job1, FeatJob* job2) {
    if (!workspace || B <= 0 || N <= 0 || M <= 0 || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    if (workspace_bytes < vpn_chamfer_workspace(B, N, M)) return VPN_E_BADARG;
    const int forced = chamfer_mode();
    if (!(forced == 6 || (forced == 0 && (long)N * M >= 512L * 512L))) return VPN_E_BADARG;   // the scan would not be mode 6
    float* ws = (float*)workspace;
    const MfmaWs w2 = mfma_carve(ws, B, M, N);
    const MfmaWs w1 = mfma_carve(ws + mfma_ws_floats(B, M, N), B, N, M);
    *job2 = FeatJob{p2, M, w2.Ntp, feat_split(w2.Ntp), w2.F, w2.nmax, w2.H, 1};
    *job1 = FeatJob{p1, N, w1.Ntp, feat_split(w1.Ntp), w1.F, w1.nmax, w1.H, 1};
    return 0;
}

// the per-workgroup sums the fp16 scan (modes 6 / 7) leaves in the workspace: sums1 [B][*g1] over dist1 (queries p1),
// sums2 [B][*g2] over dist2.  Same validity rule as chamfer_feat_jobs.
int chamfer_wgsums(const void* workspace, size_t workspace_bytes, int B, int N, int M, const float** sums1, int* g1,
                   const float** sums2, int* g2) {
    if (!workspace || B <= 0 || N <= 0 || M <= 0 || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    if (workspace_bytes < vpn_chamfer_workspace(B, N, M)) return VPN_E_BADARG;
    const int forced = chamfer_mode();
    if (!(forced == 6 || (forced == 0 && (long)N * M >= 512L * 512L))) return VPN_E_BADARG;
    float* ws = (float*)const_cast<void*>(workspace);
    const MfmaWs w2 = mfma_carve(ws, B, M, N);
    const MfmaWs w1 = mfma_carve(ws + mfma_ws_floats(B, M, N), B, N, M);
    const int qpw = 32 * CM_WAVES16;
    *sums1 = w2.wgsum; *g1 = (N + qpw - 1) / qpw;
    *sums2 = w1.wgsum; *g2 = (M + qpw - 1) / qpw;
    return 0;
}

struct CloudWs { float* sorted; int32_t* perm; float* boxes; int C; };

static inline int chunks_of(int N) { return (N + CP_CHUNK - 1) / CP_CHUNK; }
static inline size_t cloud_ws_floats(int B, int N) { return (size_t)B * ((size_t)N * 4 + (size_t)chunks_of(N) * 6); }

static CloudWs carve(float*& cur, int B, int N) {
    CloudWs w;
    w.C = chunks_of(N);
    w.sorted = cur; cur += (size_t)B * N * 3;
    w.perm = reinterpret_cast<int32_t*>(cur); cur += (size_t)B * N;
    w.boxes = cur; cur += (size_t)B * w.C * 6;
    return w;
}

static int pruned_nn(const CloudWs& q, const CloudWs& t, int B, int Nq, int Nt, float* d, int32_t* idx, hipStream_t s) {
    dim3 grid((Nq + 63) / 64, B);
    VPN_LAUNCH(chamfer_nn_pruned_kernel<1>, grid, dim3(64), 0, s, q.sorted, q.perm, t.sorted, t.perm, t.boxes,
                       Nq, Nt, t.C, d, idx);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// 0: automatic, 1: brute force, 2: box-pruned, 3: bf16 matrix-pipe filter, 4: fp32-MFMA filter
// (VPN_CHAMFER_MODE=brute|pruned|mfma|mfma32 overrides the automatic choice)
// mode 5: both clouds Morton-sorted (one launch: planes, rows, permutation, block boxes) -> pruned filtered scans
// -> fix-up of the undecided queries on the sorted planes (one launch)
static int mfma_sorted_both(const float* p1, const float* p2, int B, int N, int M, float* ws, float* d1, int32_t* i1,
                            float* d2, int32_t* i2, hipStream_t s) {
    const MfmaWs w2 = mfma_carve(ws, B, M, N);
    const MfmaWs w1 = mfma_carve(ws + mfma_ws_floats(B, M, N), B, N, M);
    const SortJob s2{p2, M, w2.Ntp, w2.F, w2.nmax, w2.H, w2.undecided, w2.perm, w2.boxes};
    const SortJob s1{p1, N, w1.Ntp, w1.F, w1.nmax, w1.H, w1.undecided, w1.perm, w1.boxes};
    VPN_LAUNCH(chamfer_sortfeat_kernel, dim3(2 * B), dim3(CS_THREADS), 0, s, s2, s1, B);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    for (int dir = 0; dir < 2; ++dir) {
        const MfmaWs& wq = dir ? w2 : w1;             // the query cloud's own sorted planes
        const MfmaWs& wt = dir ? w1 : w2;
        const int Nq = dir ? M : N, Nt = dir ? N : M;
        const int gx = (Nq + 127) / 128;
        VPN_LAUNCH(chamfer_nn_mfma_pruned_kernel, dim3(gx * B), dim3(CM_BLOCK), 0, s, wq.F, wq.perm, wq.Ntp, wt.F, wt.H,
                   wt.nmax, wt.perm, wt.boxes, Nq, Nt, wt.Ntp, gx, dir ? d2 : d1, dir ? i2 : i1, wt.undecided, B);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    const FixJob x1{w2.F, N, M, w2.Ntp, d1, i1, w2.undecided, w2.perm};
    const FixJob x2{w1.F, M, N, w1.Ntp, d2, i2, w1.undecided, w1.perm};
    VPN_LAUNCH(chamfer_fixup_kernel, dim3(2 * B * CF_GROUPS), dim3(CF_THREADS), 0, s, x1, x2, B);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static int chamfer_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("VPN_CHAMFER_MODE");
        mode = !e ? 0 : (e[0] == 'b' ? 1 : (e[0] == 'p' ? 2 : (e[0] == 'm' ? (strstr(e, "32") ? 4 : (strstr(e, "16") ? 6 : 3)) : (e[0] == 's' ? 5 : 0))));
    }
    return mode;
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
    VPN_LAUNCH(chamfer_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dist1, dist2, N, M, w1, w2,
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
    if (N <= CB_MAX_POINTS && M <= CB_MAX_POINTS) {      // per-sample LDS accumulation, no global atomics
        if (grad_p1) {
            int rc = launch_bwd_lds(p1, p2, dist1, idx1, dist2, idx2, grad_loss_b, B, N, M, w1, w2, grad_p1,
                                    (hipStream_t)stream);
            if (rc) return rc;
        }
        if (grad_p2) {
            int rc = launch_bwd_lds(p2, p1, dist2, idx2, dist1, idx1, grad_loss_b, B, M, N, w2, w1, grad_p2,
                                    (hipStream_t)stream);
            if (rc) return rc;
        }
        return 0;
    }
    int mx = N > M ? N : M;
    dim3 grid((mx + 255) / 256, B);
    VPN_LAUNCH(chamfer_bwd_direct_kernel, grid, dim3(256), 0, (hipStream_t)stream, p1, p2, dist1, idx1,
                       dist2, idx2, grad_loss_b, N, M, w1, w2, grad_p1, grad_p2);
    VPN_LAUNCH_CHECK();
    VPN_LAUNCH(chamfer_bwd_scatter_kernel, grid, dim3(256), 0, (hipStream_t)stream, p1, p2, dist1, idx1,
                       dist2, idx2, grad_loss_b, N, M, w1, w2, grad_p1, grad_p2);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t vpn_chamfer_workspace(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    const size_t pruned = cloud_ws_floats(B, N) + cloud_ws_floats(B, M);
    const size_t mfma = mfma_ws_floats(B, M, N) + mfma_ws_floats(B, N, M);
    return (pruned > mfma ? pruned : mfma) * sizeof(float);
}

// Both directions with a caller-provided workspace: clouds are Morton-sorted once and both scans are
// pruned.  mode: 0 automatic, 1 brute force, 2 box-pruned, 3 bf16 matrix-pipe filter, 4 fp32-MFMA filter
// (all bit-identical).
extern "C" int vpn_chamfer_fwd_ws(const float* p1, const float* p2, int B, int N, int M, float* dist1, int32_t* idx1,
                                  float* dist2, int32_t* idx2, void* workspace, size_t workspace_bytes, int mode,
                                  void* stream) {
    if (!p1 || !p2 || !dist1 || !idx1 || !dist2 || !idx2) return VPN_E_BADARG;
    if (B <= 0 || N <= 0 || M <= 0) return VPN_E_BADARG;
    if (B > 65535 || (long long)B * N > 0x7fffffffLL || (long long)B * M > 0x7fffffffLL) return VPN_E_TOOBIG;
    // the filtered scans trust the workspace layout (unguarded tile fetches into its slack): a short or misaligned
    // workspace is rejected here instead of faulting on the device
    if (workspace && (workspace_bytes < vpn_chamfer_workspace(B, N, M) || ((uintptr_t)workspace & 15) != 0))
        return VPN_E_BADARG;
    if (mode == 0) mode = chamfer_mode();
    // automatic: the MFMA-filtered scan for large clouds (measured 1.3x the brute-force scan at C3), brute force
    // for small ones or without a workspace; the box-pruned scan stays opt-in
    if (mode == 0) mode = (workspace && (long)N * M >= 512L * 512L) ? 6 : 1;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1 || !workspace) {
        int rc = nn_dispatch(p1, p2, B, N, M, dist1, idx1, s);
        if (rc) return rc;
        return nn_dispatch(p2, p1, B, M, N, dist2, idx2, s);
    }
    if (mode == 3 || mode == 4 || mode == 6 || mode == 7) {
        return mfma_both(p1, p2, B, N, M, (float*)workspace, dist1, idx1, dist2, idx2, mode == 4 ? 0 : (mode == 3 ? 1 : 2), s,
                         mode == 7);
    }
    if (mode == 5) return mfma_sorted_both(p1, p2, B, N, M, (float*)workspace, dist1, idx1, dist2, idx2, s);
    float* cur = (float*)workspace;
    CloudWs w1 = carve(cur, B, N), w2 = carve(cur, B, M);
    VPN_LAUNCH(cloud_sort_kernel, dim3(B), dim3(1024), 0, s, p1, N, w1.sorted, w1.perm, w1.boxes, w1.C);
    VPN_LAUNCH_CHECK();
    VPN_LAUNCH(cloud_sort_kernel, dim3(B), dim3(1024), 0, s, p2, M, w2.sorted, w2.perm, w2.boxes, w2.C);
    VPN_LAUNCH_CHECK();
    int rc = pruned_nn(w1, w2, B, N, M, dist1, idx1, s);
    if (rc) return rc;
    return pruned_nn(w2, w1, B, M, N, dist2, idx2, s);
}

// vpn_chamfer_fwd_ws of the training step with the raster's tile order as a rider of the scan launch (modes 0 / 6 / 7
// where the fp16 filter is taken: vpn_hotpath_fused_features).  records: as written by vpn_hotpath_sample_fwd earlier
// on this stream; the tile masks inside them and tile_order [B][tiles] uint16 are written here.
extern "C" size_t vpn_raster_order_size(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t ntile = (size_t)((W + R_TW - 1) / R_TW) * ((H + R_TH - 1) / R_TH);
    return 2 * (size_t)B * ntile * sizeof(TileEntry);              // entries by launch rank + the same by tile (scratch of the rider)
}

extern "C" int vpn_hotpath_chamfer_fwd(const float* p1, const float* p2, int B, int N, int M, float* dist1, int32_t* idx1,
                                       float* dist2, int32_t* idx2, void* workspace, size_t workspace_bytes, int mode,
                                       void* records, int K, int H, int W, void* tile_order, void* stream) {
    if (!p1 || !p2 || !dist1 || !idx1 || !dist2 || !idx2 || !workspace) return VPN_E_BADARG;
    if (B <= 0 || N <= 0 || M <= 0) return VPN_E_BADARG;
    if (B > 65535 || (long long)B * N > 0x7fffffffLL || (long long)B * M > 0x7fffffffLL) return VPN_E_TOOBIG;
    if (workspace_bytes < vpn_chamfer_workspace(B, N, M) || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    if (mode == 0) mode = chamfer_mode();
    if (mode == 0) mode = ((long)N * M >= 512L * 512L) ? 6 : 1;
    if (mode != 6 && mode != 7) return VPN_E_BADARG;                          // the rider lives in the fp16 scan only
    RasterOrderJob oj;
    const RasterOrderJob* rider = nullptr;
    if (tile_order) {
        if (!records || K <= 0 || K > VPN_MAX_PRIMS || H <= 0 || W <= 0 || ((uintptr_t)records & 15) != 0) return VPN_E_BADARG;
        oj.rec = (const float4*)records;
        oj.B = B; oj.K = K; oj.H = H; oj.W = W;
        oj.tiles_x = (W + R_TW - 1) / R_TW;
        oj.ntile = oj.tiles_x * ((H + R_TH - 1) / R_TH);
        oj.masks = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(records) + (size_t)B * K * R_REC * sizeof(float4));
        oj.entries = (TileEntry*)tile_order;
        if (((uintptr_t)tile_order & 15) != 0) return VPN_E_BADARG;
        if (K > R_ORDER_MAX_PRIMS || oj.ntile > R_ORDER_MAX_TILES || raster_order_scratch(K, oj.ntile) > 2 * CM_TILE16 * CM_ROWB) return VPN_E_TOOBIG;
        rider = &oj;
    }
    return mfma_both(p1, p2, B, N, M, (float*)workspace, dist1, idx1, dist2, idx2, 2, (hipStream_t)stream, mode == 7, rider);
}
