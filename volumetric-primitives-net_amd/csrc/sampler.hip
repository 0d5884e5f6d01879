// sampler.hip — surface-point sampler of volumetric primitives, forward and analytic
// backward, one launch for the whole [B,K] primitive tensor (gfx950).
//
// Replaces (reference file:line): Sampling.sphere_sampling / cuboid_sampling
// (modules/sampling/sampling.py:11-37), reparameterization + spherical_to_cartesian
// (sphere.py:22-43), cuboid quotas / reparameterization / face snapping
// (cuboid.py:30-101), transform_points (modules/transform/transform.py:6-9) and the
// per-primitive Python loop + torch.cat of sample_predict_points (train.py:105-120).
//
// Layout: one workgroup per (primitive k, sample b); the primitive's pose (R from q),
// volume and face quotas are computed once by lane 0 and staged in LDS; lanes stride
// over the n points.  Backward: dL/d(v,q,t) only needs G = sum_n g_n c_n^T (3x3) and
// sum_n g_n, reduced wave-shuffle -> LDS -> one chain-rule pass by lane 0.
#include "vpn_raster_common.h"
#include "vpn_chamfer_feat.h"

namespace vpn {

constexpr int SAMP_BLOCK = 256;

struct PrimLds {
    Pose pose;
    float v[3];
    int cum[7];   // cuboid: prefix sums of the face quotas (cum[0] = 0, cum[6] = n)
    int kind;
};

// face quotas of cuboid.py:30-53: round-half-even(n * area_f / total) for faces 0..4,
// remainder to face 5.  Same fp32 operation order as the reference.
__device__ inline void cuboid_quota(const float v[3], int n, int cum[7]) {
    float w = v[0], h = v[1], d = v[2];
    float hd = h * d, dw = d * w, wh = w * h;
    float total = (hd + dw + wh) * 2.0f;
    float area[6] = {hd, hd, dw, dw, wh, wh};
    int acc = 0;
    cum[0] = 0;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        float weight = area[f] / total;
        int c = (int)rintf((float)n * weight);
        acc += c;
        cum[f + 1] = acc;
    }
    cum[6] = n;
}

// canonical coefficient c with p_c = c * v  (unit sphere point, or unit box point snapped to a face)
__device__ inline void canonical_coeff(const PrimLds& P, int p, const float u[3], float c[3]) {
    if (P.kind == VPN_SPHERE) {
        // sphere.py:26, :38-40: elev = -acos(w) + pi/2 with w = 1 - 2 u0, then sin(elev) and cos(elev): in closed form
        // sin(elev) = w and cos(elev) = sqrt(1 - w^2) = 2 sqrt(u0 (1 - u0)) -- no acos, no sin, no cos (three of the five
        // transcendental calls of a point: the kernel is bound by instruction issue).  The reference's own chain carries
        // the rounding of pi/2 - acos(w) (6e-8 absolute in the angle); the closed form is within that of it, i.e. 1e-7
        // of a coefficient that is then scaled by the primitive's extent (the parity bar is 1e-4).
        const float w = 1.0f - 2.0f * u[0];
        const float ce = 2.0f * sqrtf(u[0] * (1.0f - u[0]));
        const float azim = u[1] * 2.0f * VPN_PI;                    // sphere.py:27
        float sa, ca;
        sincosf(azim, &sa, &ca);                                    // one range reduction for both
        c[0] = ce * sa;                                             // sphere.py:38-40
        c[1] = w;
        c[2] = ce * ca;
    } else {
        c[0] = -1.0f + 2.0f * u[0];                                 // cuboid.py:66
        c[1] = -1.0f + 2.0f * u[1];
        c[2] = -1.0f + 2.0f * u[2];
        // cuboid.py:88-99: index range -> face; later faces overwrite earlier ones only if
        // ranges overlap, which happens when a quota is negative; ranges here are disjoint.
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            if (p >= P.cum[f] && p < P.cum[f + 1]) c[f >> 1] = (f & 1) ? -1.0f : 1.0f;
        }
    }
}

__device__ inline void load_prim(PrimLds& P, const float* prm, int kind, int n) {
    P.pose = make_pose(prm[3], prm[4], prm[5], prm[6]);
    P.v[0] = prm[0]; P.v[1] = prm[1]; P.v[2] = prm[2];
    // the host binding rejects unknown kinds; device code treats anything that is not a sphere as a cuboid so that
    // no lane ever reads uninitialised quotas
    P.kind = kind == VPN_SPHERE ? VPN_SPHERE : VPN_CUBOID;
    if (P.kind == VPN_CUBOID) cuboid_quota(P.v, n, P.cum);
}

// the same with the pose the forward launch saved in the raster record (float4 R_FIN + 3 .. + 6): no sin / cos on the critical path
__device__ inline void load_prim_saved(PrimLds& P, const float* prm, int kind, int n, const float4* rk) {
    const float4 p0 = rk[R_FIN + 3], p1 = rk[R_FIN + 4], p2 = rk[R_FIN + 5], p3 = rk[R_FIN + 6];
    Pose& S = P.pose;
    S.R.m[0][0] = p0.x; S.R.m[0][1] = p0.y; S.R.m[0][2] = p0.z; S.R.m[1][0] = p0.w; S.R.m[1][1] = p1.x; S.R.m[1][2] = p1.y;
    S.R.m[2][0] = p1.z; S.R.m[2][1] = p1.w; S.R.m[2][2] = p2.x; S.x = p2.y; S.y = p2.z; S.z = p2.w;
    S.w = p3.x; S.sh = p3.y; S.ch = p3.z; S.inv_len = p3.w;
    P.v[0] = prm[0]; P.v[1] = prm[1]; P.v[2] = prm[2];
    P.kind = kind == VPN_SPHERE ? VPN_SPHERE : VPN_CUBOID;
    if (P.kind == VPN_CUBOID) cuboid_quota(P.v, n, P.cum);
}

// one workgroup = the n points of primitive k of sample b.  `feat` (training step only): the Chamfer filter's features
// of the sampled cloud (planes, fp16 rows, the slice's max norm; slot k of the sample) are written with the points --
// they used to be written by a kernel of their own that re-read the points first.
__device__ inline void sample_wg(PrimLds& P, float* red, int b, int k,
    const float* __restrict__ params, const int32_t* __restrict__ kinds, const float* __restrict__ u,
    uint64_t seed, uint64_t sample_base, int K, int n,
    float* __restrict__ points, const RasterPrep& rp, const FeatJob* feat, const FeatJob* gt = nullptr,
    float* red_gt = nullptr) {
    const float* prm = params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
    if (threadIdx.x == 0) load_prim(P, prm, kinds[k], n);
    // slice k of the ground-truth cloud's features by the waves that would otherwise wait for the pose lane
    if (gt) feat_slice_early(*gt, b, k, 64, red_gt);
    // the training step renders the same primitives: their raster records (pose, ray coefficients, culling conic) are
    // written by this launch -- one launch less per step.  The record lane (first lane of the second wave) makes the
    // camera while the pose lane makes the pose (two chains of sin / cos side by side in front of the barrier), and
    // finishes the record from both after it, while the other waves are already sampling.
    Camera C;
    const bool rec_lane = rp.rec && threadIdx.x == 64;
    if (rec_lane) C = make_camera(rp.cam + b * 3);
    // head of the loss workspace: the arrival counter is zeroed, the Philox seed this step really uses is kept (the
    // backward reads it from there: the caller's device counter may have advanced by then); per sample: tile counter
    if (rp.zero_me && k == 0) {
        if (b == 0 && threadIdx.x >= 128 && threadIdx.x < 130) rp.zero_me[threadIdx.x - 128] = 0;
        if (b == 0 && threadIdx.x == 130) *reinterpret_cast<uint64_t*>(rp.zero_me + 2) = seed;
        if (threadIdx.x == 131) rp.zero_me[4 + 4 * b + 3] = 0;
    }
    __syncthreads();
    if (gt && threadIdx.x == 65) feat_slice_publish(*gt, b, k, 64, red_gt);
    if (rec_lane) {
        float4 r[R_REC];
        make_record_from(C, P.pose, prm, kinds[k] == VPN_SPHERE ? VPN_SPHERE : VPN_CUBOID, rp.H, rp.W, rp.sigma, r);
        float4* out = rp.rec + ((size_t)b * K + k) * R_REC;
#pragma unroll
        for (int i = 0; i < R_REC; ++i) out[i] = r[i];
    }
    const float tx = prm[7], ty = prm[8], tz = prm[9];
    const float* ub = u ? u + ((size_t)b * K + k) * n * 3 : nullptr;
    float* out = points + ((size_t)b * K + k) * n * 3;
    float nv = 0.0f;
    for (int p = threadIdx.x; p < n; p += SAMP_BLOCK) {
        float uu[3];
        if (ub) { uu[0] = ub[p * 3]; uu[1] = ub[p * 3 + 1]; uu[2] = ub[p * 3 + 2]; }
        else philox_uniform3(seed, sample_base + (uint64_t)b, (uint32_t)k, (uint32_t)p, uu);
        float c[3];
        canonical_coeff(P, p, uu, c);
        float px, py, pz;
        {
            // every product and sum rounded separately, like the reference's element-wise ops (rotate.py:22-23,
            // translate.py:8) -- and so that this kernel and sample_feat_fwd_kernel give the same bits whatever the
            // compiler would contract in either
#pragma clang fp contract(off)
            const float x = c[0] * P.v[0], y = c[1] * P.v[1], z = c[2] * P.v[2];
            const Mat3& R = P.pose.R;
            px = (R.m[0][0] * x + R.m[0][1] * y + R.m[0][2] * z) + tx;
            py = (R.m[1][0] * x + R.m[1][1] * y + R.m[1][2] * z) + ty;
            pz = (R.m[2][0] * x + R.m[2][1] * y + R.m[2][2] * z) + tz;
        }
        st3(out + p * 3, px, py, pz);
        if (feat) nv = fmaxf(nv, feat_point(*feat, b, k * n + p, px, py, pz));
    }
    if (feat) {
        // the last primitive's workgroup also writes the padding rows [N, Np) of the sample (fewer than 64)
        if (k == K - 1) for (int j = K * n + (int)threadIdx.x; j < feat->Np; j += SAMP_BLOCK) feat_point(*feat, b, j, 0.f, 0.f, 0.f);
        feat_finish_slice(*feat, b, k, nv, red);
    }
}

__global__ __launch_bounds__(SAMP_BLOCK) void sample_fwd_kernel(
    const float* __restrict__ params, const int32_t* __restrict__ kinds, const float* __restrict__ u,
    uint64_t seed, const uint64_t* __restrict__ seed_dev, uint64_t sample_base, int K, int n,
    float* __restrict__ points, const RasterPrep rp) {
    __shared__ PrimLds P;
    if (seed_dev) seed += *seed_dev;
    sample_wg(P, nullptr, blockIdx.y, blockIdx.x, params, kinds, u, seed, sample_base, K, n, points, rp, nullptr);
}

// record workgroups at the head of sample_feat_fwd_kernel's grid: a multiple of 8, so that the sampler workgroups behind
// them keep their XCD (feat_decode)
__host__ __device__ inline int rec_workgroups(int BK) { return ((BK + SAMP_BLOCK - 1) / SAMP_BLOCK + 7) & ~7; }

// Forward launch of the training step with the Chamfer features inside: 1-D grid of B * K workgroups decoded like the
// feature kernel's (sample b on XCD b / (B/8)): workgroup (b, k) samples primitive k (also the slot of its max norm:
// K <= CFEAT_SLOTS) and then converts slice k of the K slices of the ground-truth cloud (64 points at C3).  The
// ground-truth slices used to be workgroups of their own: 2560 workgroups of 4 waves are 1.25 rounds of the 8192
// resident wave slots, 2048 are exactly one.
__global__ __launch_bounds__(SAMP_BLOCK) void sample_feat_fwd_kernel(
    const float* __restrict__ params, const int32_t* __restrict__ kinds, const float* __restrict__ u,
    uint64_t seed, const uint64_t* __restrict__ seed_dev, uint64_t sample_base, int B, int K, int n,
    float* __restrict__ points, const RasterPrep rp, const FeatJob pred, const FeatJob gt) {
    __shared__ PrimLds P;
    __shared__ float red[SAMP_BLOCK / 64];
    static_assert(SAMP_BLOCK == CFEAT_THREADS, "feature slices are written by sampler-sized workgroups");
    if (seed_dev) seed += *seed_dev;
    // The raster records of the step's primitives: ~1000-1400 dependent instructions each (camera, pose, culling conic or
    // silhouette hexagon).  One lane per record in workgroups of their own at the HEAD of the grid (dispatched first, done
    // long before the launch is) -- as the record lane of every sampler workgroup they were that workgroup's critical
    // path (2.6 us of the launch at C3).
    const int nrec = rp.rec ? rec_workgroups(B * K) : 0;
    if ((int)blockIdx.x < nrec) {
        const int bk = blockIdx.x * SAMP_BLOCK + threadIdx.x;
        if (bk < B * K) {
            const int rb = bk / K, rk = bk - rb * K;
            const float* prm = params + (size_t)bk * VPN_PARAM_STRIDE;
            const Camera C = make_camera(rp.cam + rb * 3);
            const Pose Pq = make_pose(prm[3], prm[4], prm[5], prm[6]);
            float4* out = rp.rec + (size_t)bk * R_REC;
            make_record_put(C, Pq, prm, kinds[rk] == VPN_SPHERE ? VPN_SPHERE : VPN_CUBOID, rp.H, rp.W, rp.sigma,
                            [&](int i, const float4 v) { out[i] = v; });
        }
        return;
    }
    RasterPrep rq = rp;
    rq.rec = nullptr;                                  // the sampler workgroups keep the counters and the seed only
    int b, sy;
    feat_decode((int)blockIdx.x - nrec, B, b, sy);
    __shared__ float red_gt[SAMP_BLOCK / 64];
    sample_wg(P, red, b, sy, params, kinds, u, seed, sample_base, K, n, points, rq, &pred, &gt, red_gt);
}

__global__ __launch_bounds__(SAMP_BLOCK) void sample_bwd_kernel(
    const float* __restrict__ params, const int32_t* __restrict__ kinds, const float* __restrict__ u,
    uint64_t seed, const uint64_t* __restrict__ seed_dev, uint64_t sample_base, int K, int n,
    const float* __restrict__ grad_points, float* __restrict__ grad_params) {
    __shared__ PrimLds P;
    if (seed_dev) seed += *seed_dev;
    __shared__ float red[SAMP_BLOCK / 64][12];
    const int k = blockIdx.x, b = blockIdx.y;
    const float* prm = params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
    if (threadIdx.x == 0) load_prim(P, prm, kinds[k], n);
    __syncthreads();
    const float* ub = u ? u + ((size_t)b * K + k) * n * 3 : nullptr;
    const float* gp = grad_points + ((size_t)b * K + k) * n * 3;
    // acc[0..8] = G[b][a] = sum g_b c_a ; acc[9..11] = sum g
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
    for (int p = threadIdx.x; p < n; p += SAMP_BLOCK) {
        float uu[3];
        if (ub) { uu[0] = ub[p * 3]; uu[1] = ub[p * 3 + 1]; uu[2] = ub[p * 3 + 2]; }
        else philox_uniform3(seed, sample_base + (uint64_t)b, (uint32_t)k, (uint32_t)p, uu);
        float c[3];
        canonical_coeff(P, p, uu, c);
        float g[3] = {gp[p * 3], gp[p * 3 + 1], gp[p * 3 + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[r * 3 + a] += g[r] * c[a];
            acc[9 + r] += g[r];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float G[3][3], gt[3];
        for (int r = 0; r < 3; ++r) {
            for (int a = 0; a < 3; ++a) {
                float s = 0.0f;
                for (int w = 0; w < SAMP_BLOCK / 64; ++w) s += red[w][r * 3 + a];
                G[r][a] = s;
            }
            float s = 0.0f;
            for (int w = 0; w < SAMP_BLOCK / 64; ++w) s += red[w][9 + r];
            gt[r] = s;
        }
        const Mat3& R = P.pose.R;
        float gR[3][3], gv[3], gq[4];
        for (int a = 0; a < 3; ++a) {
            gv[a] = R.m[0][a] * G[0][a] + R.m[1][a] * G[1][a] + R.m[2][a] * G[2][a];   // p = R (c*v) + t
            for (int r = 0; r < 3; ++r) gR[r][a] = G[r][a] * P.v[a];
        }
        pose_backward(P.pose, prm[3], prm[4], prm[5], gR, gq);
        float* o = grad_params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
        o[0] = gv[0]; o[1] = gv[1]; o[2] = gv[2];
        o[3] = gq[0]; o[4] = gq[1]; o[5] = gq[2]; o[6] = gq[3];
        o[7] = gt[0]; o[8] = gt[1]; o[9] = gt[2];
    }
}

// ---- Chamfer backward + sampler backward in one pass (the hot-path step; train.py:243-262 backward)
// The sampler's backward needs of the point gradients only G = sum_p g_p c_p^T and sum_p g_p per primitive, so the
// gradient of ChamferDistanceLoss (chamfer_distance.py:14-30) never has to exist per point: one workgroup per
// (primitive, sample) adds the direct term of its own points (their nearest GT point) and the scatter term of the
// GT points whose nearest predicted point is one of its own, each weighted by the canonical coefficient c_p that
// the forward pass used (recomputed from the same uniforms / Philox counter).  Fixed summation order: bitwise
// reproducible, unlike the LDS-atomic scatter of chamfer_bwd_lds_kernel.  Replaces that kernel (one workgroup
// per sample) plus sample_bwd_kernel and the [B,N,3] gradient between them.
// SCB_BLOCK lanes per workgroup.  128 (two waves) is the default: 2048 workgroups of four waves are 1.33 rounds of the 6144 wave
// slots this kernel's registers leave, 4096 waves are one round -- the kernel is a chain of latencies, and a wave that
// does twice the work is far shorter than two rounds of waves.
#ifndef SCB_BLOCK
#define SCB_BLOCK 128
#endif
// The other loss terms of the reference's training step (train.py:243-262) that reach the sampled points or the
// primitive centres; every coefficient is multiplied by the upstream gradient of the total in the kernel.
struct StepExtras {
    // EMD (train.py:193-195): w_emd * mean_{b,i} sqrt(dist_i), dist_i = |p_i - gt[assign_i]|^2 from vpn_emd_fwd(points, gt)
    const float* emd_dist = nullptr; const int32_t* emd_assign = nullptr; float emd_coef = 0.f;   // w_emd / (B N)
    // VP-diversity (vp_diverse.py:15-17): Chamfer(centres [B,K,3] = the translations, gt; w1 = 0.5, w2 = 1.0)
    const float* dv_dist1 = nullptr; const int32_t* dv_idx1 = nullptr;                            // [B,K]
    const float* dv_dist2 = nullptr; const int32_t* dv_idx2 = nullptr;                            // [B,M]
    float dv_c1 = 0.f, dv_c2 = 0.f;                                                               // w_div * 0.5 / (K B), w_div * 1.0 / (M B)
    // object-centred Chamfer (train.py:158-161) of canon = cn_mat_b p against cn_gt [B,Mc,3]
    const float* cn_points = nullptr; const float* cn_gt = nullptr; const float* cn_mat = nullptr;   // [B,N,3], [B,Mc,3], [B,9]
    const float* cn_dist1 = nullptr; const int32_t* cn_idx1 = nullptr; const float* cn_dist2 = nullptr; const int32_t* cn_idx2 = nullptr;
    float cn_c1 = 0.f, cn_c2 = 0.f; int cn_M = 0;                                                 // w_can * cd_w1 / (N B), w_can * cd_w2 / (Mc B)
};

template <bool EXTRAS>
__global__ __launch_bounds__(SCB_BLOCK) void sample_chamfer_bwd_kernel(
    const float* __restrict__ params, const int32_t* __restrict__ kinds, const float* __restrict__ u,
    uint64_t seed, const uint64_t* __restrict__ seed_dev, uint64_t sample_base, int K, int n,
    const float* __restrict__ points, const float* __restrict__ gt, int M, const float* __restrict__ dist1, const int32_t* __restrict__ idx1,
    const float* __restrict__ dist2, const int32_t* __restrict__ idx2, const float* __restrict__ grad_loss_b,
    float w1, float w2, float* __restrict__ grad_params, const RasterFinish rf, const StepExtras ex) {
    __shared__ PrimLds P;
    __shared__ float red[SCB_BLOCK / 64][12];
    __shared__ float fin[SCB_BLOCK / 64][12];
    __shared__ float rgrad[10];
    if (seed_dev) seed += *seed_dev;
    const int k = blockIdx.x, b = blockIdx.y, N = K * n;
    const float* prm = params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
    // The raster's finishing step for the same primitive (its gradient partials were written by the forward launch of
    // the training step) rides in this launch: every wave gathers a quarter of the tiles FIRST -- those loads depend on
    // nothing, their latency hides behind everything below -- and thread 64 applies the chain rule at the end while
    // thread 0 does the sampler's.
    if (threadIdx.x == 0) {                                             // first: everybody waits for it at the barrier
        if (rf.rec) load_prim_saved(P, prm, kinds[k], n, rf.rec + ((size_t)b * K + k) * R_REC);
        else load_prim(P, prm, kinds[k], n);
    }
    if (rf.partial) {
        float fv[16];
        raster_finish_gather(b * K + k, K, rf.ntile, rf.words, rf.masks, rf.partial, (int)(threadIdx.x >> 6) * 64, SCB_BLOCK, fv);
        const float tot = wave_reduce16_swap(fv);
        if ((threadIdx.x & 3) == 0 && (threadIdx.x & 63) < 48) fin[threadIdx.x >> 6][(threadIdx.x & 63) >> 2] = tot;
    }
    __syncthreads();
    const float* ub = u ? u + ((size_t)b * K + k) * n * 3 : nullptr;
    const float* A = points + (size_t)b * N * 3;
    const float* G2 = gt + (size_t)b * M * 3;
    const float gl = grad_loss_b ? grad_loss_b[b] : rf.scale[0];       // NULL (hot path): the upstream gradient of the total itself
    const float ca = gl * w1 / (float)N, cb = gl * w2 / (float)M;       // means over N and M (chamfer_distance.py:25-28)
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
    // The canonical coefficient c of a point (p = R (c * v) + t) is RECOVERED from the point the forward pass wrote,
    // c = R^T (p - t) / v: 12 instructions instead of redrawing the Philox uniforms and redoing acos / sin / cos
    // (~300: two thirds of this kernel's instructions).  It differs from the forward's c by rounding only (1e-7
    // relative; the gradient is a 1e-4 contract).  The error of a recovered c_a is ~2^-24 |p| / |v_a| (p - t cancels):
    // a THIN primitive (smallest extent below 1e-3 of the size of the numbers it is subtracted from; the network's
    // range, vpnet_one_resnet.py:71,84, stays 10x above that) falls back to the exact recomputation from the uniforms /
    // the Philox counter, uniformly for the workgroup.
    const float tx = prm[7], ty = prm[8], tz = prm[9];
    const float vmin = fminf(fminf(fabsf(P.v[0]), fabsf(P.v[1])), fabsf(P.v[2]));
    const float big = fmaxf(fmaxf(fmaxf(fabsf(tx), fabsf(ty)), fabsf(tz)), fmaxf(fmaxf(fabsf(P.v[0]), fabsf(P.v[1])), fabsf(P.v[2])));
    const bool recover = vmin >= 1.0e-3f * big && vmin > 1e-20f;
    const float iv0 = recover ? 1.0f / P.v[0] : 0.0f, iv1 = recover ? 1.0f / P.v[1] : 0.0f, iv2 = recover ? 1.0f / P.v[2] : 0.0f;
    auto add = [&](int pl, float gx, float gy, float gz) {              // point pl of this primitive gets gradient g
        float c[3];
        if (recover) {
            const int i = k * n + pl;
            const F3 a = ld3(A + i * 3);
            const float dx = a.x - tx, dy = a.y - ty, dz = a.z - tz;
            const Mat3& R = P.pose.R;
            c[0] = (R.m[0][0] * dx + R.m[1][0] * dy + R.m[2][0] * dz) * iv0;
            c[1] = (R.m[0][1] * dx + R.m[1][1] * dy + R.m[2][1] * dz) * iv1;
            c[2] = (R.m[0][2] * dx + R.m[1][2] * dy + R.m[2][2] * dz) * iv2;
        } else {
            float uu[3];
            if (ub) { uu[0] = ub[pl * 3]; uu[1] = ub[pl * 3 + 1]; uu[2] = ub[pl * 3 + 2]; }
            else philox_uniform3(seed, sample_base + (uint64_t)b, (uint32_t)k, (uint32_t)pl, uu);
            canonical_coeff(P, pl, uu, c);
        }
        const float g[3] = {gx, gy, gz};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[r * 3 + a] += g[r] * c[a];
            acc[9 + r] += g[r];
        }
    };
    // object-centred Chamfer: canon = S p with S = cn_mat_b (3x3, the camera rotations times dist); a gradient g of a
    // canonical point is S^T g for the view-centred point it came from
    // (named scalars: as an array that is only filled on one branch the matrix went to scratch memory)
    const bool cn_on = EXTRAS && ex.cn_points;
    const float* Sm = ex.cn_mat + b * 9;
    const float S0 = cn_on ? Sm[0] : 0.f, S1 = cn_on ? Sm[1] : 0.f, S2 = cn_on ? Sm[2] : 0.f, S3 = cn_on ? Sm[3] : 0.f, S4 = cn_on ? Sm[4] : 0.f,
                S5 = cn_on ? Sm[5] : 0.f, S6 = cn_on ? Sm[6] : 0.f, S7 = cn_on ? Sm[7] : 0.f, S8 = cn_on ? Sm[8] : 0.f;
    auto add_canon = [&](int pl, float gx, float gy, float gz) {
        add(pl, S0 * gx + S3 * gy + S6 * gz, S1 * gx + S4 * gy + S7 * gz, S2 * gx + S5 * gy + S8 * gz);
    };
    for (int pl = threadIdx.x; pl < n; pl += SCB_BLOCK) {              // own nearest neighbour
        const int i = k * n + pl, j = idx1[(size_t)b * N + i];
        const float coef = ca / dist1[(size_t)b * N + i];
        const F3 a = ld3(A + i * 3), g = ld3(G2 + j * 3);
        float gx = coef * (a.x - g.x), gy = coef * (a.y - g.y), gz = coef * (a.z - g.z);
        if (EXTRAS && ex.emd_dist) {
            // d mean sqrt(dist) / d p = (p - y) / sqrt(dist) / (B N)  (sqrt and mean of train.py:195 through NmDistanceGradKernel,
            // emd_cuda.cu:284-300: 2 g (p - y) with g = 1 / (2 sqrt(dist)));  dist = 0 gives NaN like the reference's chain
            const int t = ex.emd_assign[(size_t)b * N + i];
            if (t >= 0 && t < M) {
                const F3 y = ld3(G2 + t * 3);
                const float ce = (gl * ex.emd_coef) / sqrtf(ex.emd_dist[(size_t)b * N + i]);
                gx += ce * (a.x - y.x); gy += ce * (a.y - y.y); gz += ce * (a.z - y.z);
            } else {
                gx = gy = gz = __builtin_nanf("");                  // the auction left the point unassigned: say so
            }
        }
        add(pl, gx, gy, gz);
        if (EXTRAS && ex.cn_points) {
            const int jc = ex.cn_idx1[(size_t)b * N + i];
            const float cc = (gl * ex.cn_c1) / ex.cn_dist1[(size_t)b * N + i];
            const F3 ac = ld3(ex.cn_points + ((size_t)b * N + i) * 3), gc = ld3(ex.cn_gt + ((size_t)b * ex.cn_M + jc) * 3);
            add_canon(pl, cc * (ac.x - gc.x), cc * (ac.y - gc.y), cc * (ac.z - gc.z));
        }
    }
    // being a GT point's nearest neighbour: about M/K of the M entries concern this primitive, spread so that almost
    // every pass over 256 entries has a lane with a match — handled in place, the heavy body (dependent loads,
    // Philox) ran M/256 times with one or two active lanes.  So each wave first compacts its matches into LDS
    // (ballot prefix: a fixed order) and then handles them densely.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ int2 match[];                                     // [SCB_BLOCK/64][cap] of (GT point, predicted point)
    const int cap = (M + SCB_BLOCK - 1) / SCB_BLOCK * 64;             // entries one wave looks at
    int2* mine = match + wave * cap;
    int cnt = 0;
    // 8 passes at a time: their loads are issued together (each pass used to wait out its own L2 round trip)
    // (training step: the VP-diversity term's scatter -- ground-truth points whose nearest CENTRE is this primitive's -- rides
    // in the same pass over the M entries; the centre of primitive k IS t_k, so its gradient goes straight into sum g)
    float dvx = 0.f, dvy = 0.f, dvz = 0.f;
    const bool dv_on = EXTRAS && ex.dv_dist1;
    const float dv_c2 = dv_on ? gl * ex.dv_c2 : 0.0f;
    for (int e0 = 0; e0 < M; e0 += 8 * SCB_BLOCK) {
        int iv[8], dvi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * SCB_BLOCK + threadIdx.x;
            iv[u] = e < M ? idx2[(size_t)b * M + e] : -1;
            dvi[u] = (dv_on && e < M) ? ex.dv_idx2[(size_t)b * M + e] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (e0 + u * SCB_BLOCK >= M) break;
            if (dv_on && dvi[u] == k) {
                const int e = e0 + u * SCB_BLOCK + (int)threadIdx.x;
                const float cc = dv_c2 / ex.dv_dist2[(size_t)b * M + e];
                const F3 g = ld3(G2 + e * 3);
                dvx += cc * (tx - g.x); dvy += cc * (ty - g.y); dvz += cc * (tz - g.z);
            }
            const bool hit = iv[u] >= k * n && iv[u] < (k + 1) * n;
            const unsigned long long m = __ballot(hit);
            if (hit) mine[cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = make_int2(e0 + u * SCB_BLOCK + (int)threadIdx.x, iv[u]);
            cnt += __builtin_popcountll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");             // the wave reads its own LDS writes below
    __builtin_amdgcn_wave_barrier();
    for (int l = lane; l < cnt; l += 64) {
        const int e = mine[l].x, i = mine[l].y;                         // the index was read during the scan: no second look-up
        const float coef = cb / dist2[(size_t)b * M + e];
        const F3 a = ld3(A + i * 3), g = ld3(G2 + e * 3);
        add(i - k * n, coef * (a.x - g.x), coef * (a.y - g.y), coef * (a.z - g.z));
    }
    if (EXTRAS && ex.cn_points) {
        // the object-centred cloud's scatter term: ground-truth points whose nearest canonical point is one of this
        // primitive's (their number is small: handled where they are found)
        const float cbc = gl * ex.cn_c2;
        for (int e = threadIdx.x; e < ex.cn_M; e += SCB_BLOCK) {
            const int i = ex.cn_idx2[(size_t)b * ex.cn_M + e];
            if (i < k * n || i >= (k + 1) * n) continue;
            const float cc = cbc / ex.cn_dist2[(size_t)b * ex.cn_M + e];
            const F3 ac = ld3(ex.cn_points + ((size_t)b * N + i) * 3), gc = ld3(ex.cn_gt + ((size_t)b * ex.cn_M + e) * 3);
            add_canon(i - k * n, cc * (ac.x - gc.x), cc * (ac.y - gc.y), cc * (ac.z - gc.z));
        }
    }
    if (dv_on) {
        // VP-diversity: the scatter part was gathered in the pass over the M entries above; the centre's own nearest neighbour
        float sx = dvx, sy = dvy, sz = dvz;
        if (threadIdx.x == 0) {
            const int j = ex.dv_idx1[b * K + k];
            const float cc = (gl * ex.dv_c1) / ex.dv_dist1[b * K + k];
            const F3 g = ld3(G2 + j * 3);
            sx += cc * (tx - g.x); sy += cc * (ty - g.y); sz += cc * (tz - g.z);
        }
        acc[9] += sx; acc[10] += sy; acc[11] += sz;
    }
    {   // 12 sums per wave with ONE transposing butterfly (17 cross-lane moves) instead of 12 x 6
        float v16[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v16[i] = i < 12 ? acc[i] : 0.0f;
        const float tot = wave_reduce16_swap(v16);                           // lane L: wave total of value L >> 2
        if ((lane & 3) == 0 && lane < 48) red[wave][lane >> 2] = tot;
    }
    __syncthreads();
    if (rf.partial && threadIdx.x == 64) {                              // raster chain rule, beside thread 0's below
        float G[12], r[10];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            float t = fin[0][i];
#pragma unroll
            for (int w = 1; w < SCB_BLOCK / 64; ++w) t += fin[w][i];
            G[i] = t;
        }
        raster_finish_chain(params, rf.rec, b * K + k, K, G, r);
        const float sc = rf.scale ? *rf.scale : 1.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) rgrad[i] = sc * r[i];
    }
    float res[10];
    if (threadIdx.x == 0) {
        float G[3][3], gt3[3];
        for (int r = 0; r < 3; ++r) {
            for (int a = 0; a < 3; ++a) {
                float s = 0.0f;
                for (int w = 0; w < SCB_BLOCK / 64; ++w) s += red[w][r * 3 + a];
                G[r][a] = s;
            }
            float s = 0.0f;
            for (int w = 0; w < SCB_BLOCK / 64; ++w) s += red[w][9 + r];
            gt3[r] = s;
        }
        const Mat3& R = P.pose.R;
        float gR[3][3], gv[3], gq[4];
        for (int a = 0; a < 3; ++a) {
            gv[a] = R.m[0][a] * G[0][a] + R.m[1][a] * G[1][a] + R.m[2][a] * G[2][a];   // p = R (c*v) + t
            for (int r = 0; r < 3; ++r) gR[r][a] = G[r][a] * P.v[a];
        }
        pose_backward(P.pose, prm[3], prm[4], prm[5], gR, gq);
        res[0] = gv[0]; res[1] = gv[1]; res[2] = gv[2]; res[3] = gq[0]; res[4] = gq[1]; res[5] = gq[2]; res[6] = gq[3];
        res[7] = gt3[0]; res[8] = gt3[1]; res[9] = gt3[2];
    }
    if (rf.partial) __syncthreads();                                    // rgrad (thread 64) -> thread 0; uniform condition
    if (threadIdx.x == 0) {
        float* o = grad_params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[i] = rf.partial ? res[i] + rgrad[i] : res[i];
    }
}

// ---- standalone transform (modules/transform/transform.py:6-9): out = R(q) p + t
constexpr int TR_BLOCK = 256;

__global__ __launch_bounds__(TR_BLOCK) void transform_fwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ q, const float* __restrict__ t, int N,
    float* __restrict__ out) {
    __shared__ Pose S;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) S = make_pose(q[b * 4], q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]);
    __syncthreads();
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (t) { tx = t[b * 3]; ty = t[b * 3 + 1]; tz = t[b * 3 + 2]; }
    const Mat3& R = S.R;
    for (int p = blockIdx.x * TR_BLOCK + threadIdx.x; p < N; p += gridDim.x * TR_BLOCK) {
        const float* pp = pts + ((size_t)b * N + p) * 3;
        float x = pp[0], y = pp[1], z = pp[2];
        float* o = out + ((size_t)b * N + p) * 3;
        float ox = R.m[0][0] * x + R.m[0][1] * y + R.m[0][2] * z;
        float oy = R.m[1][0] * x + R.m[1][1] * y + R.m[1][2] * z;
        float oz = R.m[2][0] * x + R.m[2][1] * y + R.m[2][2] * z;
        if (t) { ox += tx; oy += ty; oz += tz; }
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
}

// one workgroup per sample: grad_points = R^T g; G = sum g p^T; sum g
__global__ __launch_bounds__(TR_BLOCK) void transform_bwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ q, const float* __restrict__ gout, int N,
    float* __restrict__ gpts, float* __restrict__ gq, float* __restrict__ gt) {
    __shared__ Pose S;
    __shared__ float red[TR_BLOCK / 64][12];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) S = make_pose(q[b * 4], q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]);
    __syncthreads();
    const Mat3& R = S.R;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
    for (int p = threadIdx.x; p < N; p += TR_BLOCK) {
        const float* gg = gout + ((size_t)b * N + p) * 3;
        const float* pp = pts + ((size_t)b * N + p) * 3;
        float g[3] = {gg[0], gg[1], gg[2]};
        float c[3] = {pp[0], pp[1], pp[2]};
        if (gpts) {
            float* o = gpts + ((size_t)b * N + p) * 3;
            o[0] = R.m[0][0] * g[0] + R.m[1][0] * g[1] + R.m[2][0] * g[2];
            o[1] = R.m[0][1] * g[0] + R.m[1][1] * g[1] + R.m[2][1] * g[2];
            o[2] = R.m[0][2] * g[0] + R.m[1][2] * g[1] + R.m[2][2] * g[2];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[r * 3 + a] += g[r] * c[a];
            acc[9 + r] += g[r];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float G[3][3], st[3];
        for (int r = 0; r < 3; ++r) {
            for (int a = 0; a < 3; ++a) {
                float s = 0.0f;
                for (int w = 0; w < TR_BLOCK / 64; ++w) s += red[w][r * 3 + a];
                G[r][a] = s;
            }
            float s = 0.0f;
            for (int w = 0; w < TR_BLOCK / 64; ++w) s += red[w][9 + r];
            st[r] = s;
        }
        if (gq) {
            float o[4];
            pose_backward(S, q[b * 4], q[b * 4 + 1], q[b * 4 + 2], G, o);
            gq[b * 4] = o[0]; gq[b * 4 + 1] = o[1]; gq[b * 4 + 2] = o[2]; gq[b * 4 + 3] = o[3];
        }
        if (gt) { gt[b * 3] = st[0]; gt[b * 3 + 1] = st[1]; gt[b * 3 + 2] = st[2]; }
    }
}

// ---- primitive -> mesh vertices (row f2: modules/meshing/sphere.py:8-27, cuboid.py:8-26, meshing.py:27-46)
// The reference parses an OBJ template from disk for every (sample, primitive), scales its vertices by v,
// runs transform_points and concatenates the K meshes.  Here the templates sit on the device and one launch writes
// verts[b][offsets[k] + p] = R(q_bk) (tpl_kind(k)[p] * v_bk) + t_bk for all (b, k).  offsets [K+1] are the vertex
// offsets of the composed mesh (compose_meshes appends the primitives in order).
constexpr int MESH_BLOCK = 256;

__global__ __launch_bounds__(MESH_BLOCK) void mesh_fwd_kernel(const float* __restrict__ params,
                                                              const int32_t* __restrict__ kinds,
                                                              const int32_t* __restrict__ offsets,
                                                              const float* __restrict__ tpl_sphere,
                                                              const float* __restrict__ tpl_cuboid, int K, int Ptot,
                                                              float* __restrict__ verts) {
    __shared__ Pose S;
    const int k = blockIdx.x, b = blockIdx.y;
    const float* prm = params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
    if (threadIdx.x == 0) S = make_pose(prm[3], prm[4], prm[5], prm[6]);
    __syncthreads();
    const Mat3& R = S.R;
    const float* tpl = kinds[k] == VPN_SPHERE ? tpl_sphere : tpl_cuboid;
    const int o = offsets[k], P = offsets[k + 1] - o;
    const float v0 = prm[0], v1 = prm[1], v2 = prm[2], t0 = prm[7], t1 = prm[8], t2 = prm[9];
    float* out = verts + ((size_t)b * Ptot + o) * 3;
    for (int p = threadIdx.x; p < P; p += MESH_BLOCK) {
        const float x = tpl[p * 3] * v0, y = tpl[p * 3 + 1] * v1, z = tpl[p * 3 + 2] * v2;     // sphere.py:15
        out[p * 3 + 0] = (R.m[0][0] * x + R.m[0][1] * y + R.m[0][2] * z) + t0;
        out[p * 3 + 1] = (R.m[1][0] * x + R.m[1][1] * y + R.m[1][2] * z) + t1;
        out[p * 3 + 2] = (R.m[2][0] * x + R.m[2][1] * y + R.m[2][2] * z) + t2;
    }
}

// backward: the same chain rule as the sampler's (the canonical coefficient of a vertex is its template coordinate)
__global__ __launch_bounds__(MESH_BLOCK) void mesh_bwd_kernel(const float* __restrict__ params,
                                                              const int32_t* __restrict__ kinds,
                                                              const int32_t* __restrict__ offsets,
                                                              const float* __restrict__ tpl_sphere,
                                                              const float* __restrict__ tpl_cuboid, int K, int Ptot,
                                                              const float* __restrict__ grad_verts,
                                                              float* __restrict__ grad_params) {
    __shared__ Pose S;
    __shared__ float red[MESH_BLOCK / 64][12];
    const int k = blockIdx.x, b = blockIdx.y;
    const float* prm = params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
    if (threadIdx.x == 0) S = make_pose(prm[3], prm[4], prm[5], prm[6]);
    __syncthreads();
    const float* tpl = kinds[k] == VPN_SPHERE ? tpl_sphere : tpl_cuboid;
    const int o = offsets[k], P = offsets[k + 1] - o;
    const float* gp = grad_verts + ((size_t)b * Ptot + o) * 3;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
    for (int p = threadIdx.x; p < P; p += MESH_BLOCK) {
        const float c[3] = {tpl[p * 3], tpl[p * 3 + 1], tpl[p * 3 + 2]};
        const float g[3] = {gp[p * 3], gp[p * 3 + 1], gp[p * 3 + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[r * 3 + a] += g[r] * c[a];
            acc[9 + r] += g[r];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float G[3][3], gt[3];
        for (int r = 0; r < 3; ++r) {
            for (int a = 0; a < 3; ++a) {
                float s = 0.0f;
                for (int w = 0; w < MESH_BLOCK / 64; ++w) s += red[w][r * 3 + a];
                G[r][a] = s;
            }
            float s = 0.0f;
            for (int w = 0; w < MESH_BLOCK / 64; ++w) s += red[w][9 + r];
            gt[r] = s;
        }
        const Mat3& R = S.R;
        float gR[3][3], gv[3], gq[4];
        for (int a = 0; a < 3; ++a) {
            gv[a] = R.m[0][a] * G[0][a] + R.m[1][a] * G[1][a] + R.m[2][a] * G[2][a];   // p = R (c*v) + t
            for (int r = 0; r < 3; ++r) gR[r][a] = G[r][a] * prm[a];
        }
        pose_backward(S, prm[3], prm[4], prm[5], gR, gq);
        float* og = grad_params + ((size_t)b * K + k) * VPN_PARAM_STRIDE;
        og[0] = gv[0]; og[1] = gv[1]; og[2] = gv[2];
        og[3] = gq[0]; og[4] = gq[1]; og[5] = gq[2]; og[6] = gq[3];
        og[7] = gt[0]; og[8] = gt[1]; og[9] = gt[2];
    }
}

// ---- fused camera transforms (modules/transform/transform.py:21-73): the reference chains 3-4 rotate_points
// calls and a scale over the whole cloud; here the per-sample 3x3 is composed once and applied in one pass.
//   to_object != 0  view_to_obj_points :21-47:  p * dist <- R(-z,-e) R(y',-a) R(x,-angle) p,  y' = R(-z,e) y
//   to_object == 0  obj_to_view_points :50-73:  p / dist <- R(y', a) R(-z, e) p
// (angles in degrees / 360 = turns, the unit of refine_quaternions, rotate.py:59-72)
__device__ inline Mat3 mat_mul(const Mat3& A, const Mat3& B) {
    Mat3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}

__device__ inline Mat3 camera_matrix(float elev, float azim, float angle, int to_object) {
    const float e = elev / 360.0f, a = azim / 360.0f;
    const Mat3 Re = make_pose(0.0f, 0.0f, -1.0f, e).R;                    // :32-33 / :60-64
    const float yx = Re.m[0][1], yy = Re.m[1][1], yz = Re.m[2][1];         // R(-z,e) (0,1,0)
    if (to_object) {
        const Mat3 Rx = make_pose(1.0f, 0.0f, 0.0f, -angle / 360.0f).R;    // :25, :88-92
        const Mat3 Ra = make_pose(yx, yy, yz, -a).R;                       // :35-36
        const Mat3 Rb = make_pose(0.0f, 0.0f, -1.0f, -e).R;                // :38-39
        return mat_mul(Rb, mat_mul(Ra, Rx));
    }
    const Mat3 Ra = make_pose(yx, yy, yz, a).R;                            // :66-67
    return mat_mul(Ra, Re);
}

// transpose != 0: the backward pass (grad_points = M^T g, same scale)
__global__ __launch_bounds__(TR_BLOCK) void camera_transform_kernel(
    const float* __restrict__ pts, const float* __restrict__ dists, const float* __restrict__ elevs,
    const float* __restrict__ azims, const float* __restrict__ angles, int N, int to_object, int transpose,
    float* __restrict__ out) {
    __shared__ Mat3 S;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        Mat3 M = camera_matrix(elevs[b], azims[b], angles ? angles[b] : 0.0f, to_object);
        if (transpose) {
            Mat3 T;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.m[i][j] = M.m[j][i];
            M = T;
        }
        S = M;
    }
    __syncthreads();
    const Mat3& R = S;
    const float d = dists[b];
    for (int p = blockIdx.x * TR_BLOCK + threadIdx.x; p < N; p += gridDim.x * TR_BLOCK) {
        const float* pp = pts + ((size_t)b * N + p) * 3;
        const float x = pp[0], y = pp[1], z = pp[2];
        float ox = R.m[0][0] * x + R.m[0][1] * y + R.m[0][2] * z;
        float oy = R.m[1][0] * x + R.m[1][1] * y + R.m[1][2] * z;
        float oz = R.m[2][0] * x + R.m[2][1] * y + R.m[2][2] * z;
        if (to_object) { ox *= d; oy *= d; oz *= d; }                      // :44
        else { ox /= d; oy /= d; oz /= d; }                                // :71
        float* o = out + ((size_t)b * N + p) * 3;
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
}

// the per-sample 3x3 of the fused camera transforms, scale included: out = mat p  (backward of the object-centred Chamfer)
__global__ __launch_bounds__(64) void camera_matrix_kernel(const float* __restrict__ dists, const float* __restrict__ elevs,
                                                          const float* __restrict__ azims, const float* __restrict__ angles,
                                                          int B, int to_object, float* __restrict__ mat) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const Mat3 M = camera_matrix(elevs[b], azims[b], angles ? angles[b] : 0.0f, to_object);
    const float d = dists[b];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) mat[b * 9 + i * 3 + j] = to_object ? M.m[i][j] * d : M.m[i][j] / d;
}

// records of all (image, primitive) pairs: one lane each (also zeroes the arrival counter of the loss finalisation)
__global__ __launch_bounds__(256) void raster_prep_kernel(const float* __restrict__ params,
                                                          const int32_t* __restrict__ kinds,
                                                          const float* __restrict__ cam, int BK, int K, int H, int W,
                                                          float sigma, float4* __restrict__ rec, int* __restrict__ zero_me) {
    const int bk = blockIdx.x * 256 + threadIdx.x;
    if (zero_me && bk < 4) zero_me[bk] = 0;
    if (bk >= BK) return;
    const int b = bk / K, k = bk - b * K;
    if (zero_me && k == 0) zero_me[4 + 4 * b + 3] = 0;           // the sample's tile counter (fused finalisation)
    float4 r[R_REC];
    make_record(params + (size_t)bk * VPN_PARAM_STRIDE, kinds[k] == VPN_SPHERE ? VPN_SPHERE : VPN_CUBOID, cam, b, H, W, sigma, r);
    float4* out = rec + (size_t)bk * R_REC;
#pragma unroll
    for (int i = 0; i < R_REC; ++i) out[i] = r[i];
}


int launch_raster_prep(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H, int W,
                       float sigma, void* records, int* zero_me, hipStream_t s) {
    const int BK = B * K;
    VPN_LAUNCH(raster_prep_kernel, dim3((BK + 255) / 256), dim3(256), 0, s, params, kinds, cam, BK, K, H, W, sigma,
               (float4*)records, zero_me);
    VPN_LAUNCH_CHECK();
    return 0;
}

}  // namespace vpn

using namespace vpn;

static int launch_sample_fwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                             const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, float* points,
                             const RasterPrep& rp, void* stream) {
    if (!params || !kinds || !points) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || n <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    VPN_LAUNCH(sample_fwd_kernel, dim3(K, B), dim3(SAMP_BLOCK), 0, (hipStream_t)stream, params, kinds, u,
                       seed, seed_dev, sample_base, K, n, points, rp);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_sample_fwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                              const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, float* points,
                              void* stream) {
    return launch_sample_fwd(params, kinds, u, seed, seed_dev, sample_base, B, K, n, points, RasterPrep{}, stream);
}

// sampler forward of the training step: also writes the raster records of the same primitives (what the first launch
// of vpn_raster_total_fwd would compute) and zeroes the arrival counter at the head of loss_ws; with a Chamfer
// workspace it also writes the matrix-pipe filter's features of the sampled cloud and of gt_points into it
extern "C" int vpn_hotpath_sample_fwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                                      const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, float* points,
                                      const float* cam, int H, int W, float sigma, void* records, void* loss_ws,
                                      const float* gt_points, int M, void* chamfer_ws, size_t chamfer_ws_bytes,
                                      void* stream) {
    if (!cam || !records || H <= 0 || W <= 0 || !(sigma > 0.f) || K > VPN_MAX_PRIMS) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0) return VPN_E_BADARG;
    RasterPrep rp;
    rp.cam = cam; rp.H = H; rp.W = W; rp.sigma = sigma; rp.rec = (float4*)records; rp.zero_me = (int*)loss_ws;
    if (!chamfer_ws) return launch_sample_fwd(params, kinds, u, seed, seed_dev, sample_base, B, K, n, points, rp, stream);
    if (!params || !kinds || !points || !gt_points) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || n <= 0 || M <= 0 || K > CFEAT_SLOTS) return VPN_E_BADARG;
    if ((long long)B * K * n > 0x7fffffffLL) return VPN_E_TOOBIG;
    FeatJob pred, gt;
    int rc = chamfer_feat_jobs(chamfer_ws, chamfer_ws_bytes, B, K * n, M, points, gt_points, &pred, &gt);
    if (rc) return rc;
    pred.ysplit = K;                      // one slice (and one max-norm slot) per primitive
    gt.ysplit = K;                        // ... and the same workgroups share the ground-truth cloud
    const unsigned nrec = (unsigned)rec_workgroups(B * K);
    VPN_LAUNCH(sample_feat_fwd_kernel, dim3((unsigned)B * (unsigned)K + nrec), dim3(SAMP_BLOCK), 0, (hipStream_t)stream,
               params, kinds, u, seed, seed_dev, sample_base, B, K, n, points, rp, pred, gt);
    VPN_LAUNCH_CHECK();
    return 0;
}

// 1 if vpn_hotpath_sample_fwd can write the Chamfer features for these sizes (the automatic mode of
// vpn_chamfer_fwd_ws takes the fp16 filter and every primitive gets its own max-norm slot): the caller then runs
// vpn_chamfer_fwd_ws with mode 7 on the same workspace and stream
extern "C" int vpn_hotpath_fused_features(int B, int K, int n, int M) {
    if (B <= 0 || K <= 0 || n <= 0 || M <= 0 || K > CFEAT_SLOTS) return 0;
    FeatJob a, b;
    static float dummy[4] __attribute__((aligned(16)));
    // only the mode decision of chamfer_feat_jobs matters here: a workspace "large enough" is pretended
    return chamfer_feat_jobs(dummy, (size_t)-1, B, K * n, M, nullptr, nullptr, &a, &b) == 0 ? 1 : 0;
}

extern "C" int vpn_sample_bwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                              const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, const float* grad_points,
                              float* grad_params, void* stream) {
    if (!params || !kinds || !grad_points || !grad_params) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || n <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    VPN_LAUNCH(sample_bwd_kernel, dim3(K, B), dim3(SAMP_BLOCK), 0, (hipStream_t)stream, params, kinds, u,
                       seed, seed_dev, sample_base, K, n, grad_points, grad_params);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_transform_fwd(const float* points, const float* q, const float* t, int B, int N, float* out,
                                 void* stream) {
    if (!points || !q || !out) return VPN_E_BADARG;
    if (B <= 0 || N <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    int gx = (N + TR_BLOCK - 1) / TR_BLOCK;
    if (gx > 1024) gx = 1024;
    VPN_LAUNCH(transform_fwd_kernel, dim3(gx, B), dim3(TR_BLOCK), 0, (hipStream_t)stream, points, q, t, N,
                       out);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_transform_bwd(const float* points, const float* q, const float* grad_out, int B, int N,
                                 float* grad_points, float* grad_q, float* grad_t, void* stream) {
    if (!points || !q || !grad_out) return VPN_E_BADARG;
    if (B <= 0 || N <= 0) return VPN_E_BADARG;
    VPN_LAUNCH(transform_bwd_kernel, dim3(B), dim3(TR_BLOCK), 0, (hipStream_t)stream, points, q, grad_out,
                       N, grad_points, grad_q, grad_t);
    VPN_LAUNCH_CHECK();
    return 0;
}

static int camera_launch(const float* points, const float* dists, const float* elevs, const float* azims,
                         const float* angles, int B, int N, int to_object, int transpose, float* out, void* stream) {
    if (!points || !dists || !elevs || !azims || !out) return VPN_E_BADARG;
    if (to_object && !angles) return VPN_E_BADARG;
    if (B <= 0 || N <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    int gx = (N + TR_BLOCK - 1) / TR_BLOCK;
    if (gx > 1024) gx = 1024;
    VPN_LAUNCH(camera_transform_kernel, dim3(gx, B), dim3(TR_BLOCK), 0, (hipStream_t)stream, points, dists, elevs,
               azims, angles, N, to_object, transpose, out);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_camera_matrix(const float* dists, const float* elevs, const float* azims, const float* angles, int B,
                                 int to_object, float* mat, void* stream) {
    if (!dists || !elevs || !azims || !mat || (to_object && !angles) || B <= 0) return VPN_E_BADARG;
    VPN_LAUNCH(camera_matrix_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, dists, elevs, azims, angles, B, to_object, mat);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_camera_transform_fwd(const float* points, const float* dists, const float* elevs,
                                        const float* azims, const float* angles, int B, int N, int to_object,
                                        float* out, void* stream) {
    return camera_launch(points, dists, elevs, azims, angles, B, N, to_object, 0, out, stream);
}

extern "C" int vpn_camera_transform_bwd(const float* grad_out, const float* dists, const float* elevs,
                                        const float* azims, const float* angles, int B, int N, int to_object,
                                        float* grad_points, void* stream) {
    return camera_launch(grad_out, dists, elevs, azims, angles, B, N, to_object, 1, grad_points, stream);
}

static int launch_scb(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                      const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, const float* points,
                      const float* gt_points, int M, const float* dist1, const int32_t* idx1,
                      const float* dist2, const int32_t* idx2, const float* grad_loss_b, float w1,
                      float w2, float* grad_params, const RasterFinish& rf, void* stream, const StepExtras* ex = nullptr) {
    if (!params || !kinds || !points || !gt_points || !dist1 || !idx1 || !dist2 || !idx2 || !grad_params)
        return VPN_E_BADARG;
    if (!grad_loss_b && !rf.scale) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || n <= 0 || M <= 0) return VPN_E_BADARG;
    if (B > 65535 || (long long)K * n > 0x7fffffffLL / 3) return VPN_E_TOOBIG;
    constexpr int BLK = SCB_BLOCK;
    const size_t lds = (size_t)((M + BLK - 1) / BLK) * 64 * (BLK / 64) * sizeof(int2);  // = 8 B per GT point
    if (lds > 60 * 1024) return VPN_E_TOOBIG;                           // 7680 GT points; beyond: vpn_chamfer_bwd + vpn_sample_bwd
    if (ex)
        VPN_LAUNCH_AS("sample_chamfer_bwd_kernel<step>", sample_chamfer_bwd_kernel<true>, dim3(K, B), dim3(BLK), lds, (hipStream_t)stream, params, kinds, u, seed,
                      seed_dev, sample_base, K, n, points, gt_points, M, dist1, idx1, dist2, idx2, grad_loss_b, w1, w2, grad_params, rf, *ex);
    else
        VPN_LAUNCH_AS("sample_chamfer_bwd_kernel", sample_chamfer_bwd_kernel<false>, dim3(K, B), dim3(BLK), lds, (hipStream_t)stream, params, kinds, u, seed,
                      seed_dev, sample_base, K, n, points, gt_points, M, dist1, idx1, dist2, idx2, grad_loss_b, w1, w2, grad_params, rf, StepExtras{});
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_sample_chamfer_bwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                                      const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, const float* points,
                                      const float* gt_points, int M, const float* dist1, const int32_t* idx1,
                                      const float* dist2, const int32_t* idx2, const float* grad_loss_b, float w1,
                                      float w2, float* grad_params, void* stream) {
    return launch_scb(params, kinds, u, seed, seed_dev, sample_base, B, K, n, points, gt_points, M, dist1, idx1, dist2, idx2,
                      grad_loss_b, w1, w2, grad_params, RasterFinish{}, stream);
}

// the same launch also finishes the raster backward of the training step (vpn_raster_total_fwd wrote the partials):
// grad_params = d(Chamfer term)/d params + (*grad_total) * d(total_img)/d params
extern "C" int vpn_hotpath_bwd(const float* params, const int32_t* kinds, const float* u, uint64_t seed,
                               const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n, const float* points,
                               const float* gt_points, int M, const float* dist1, const int32_t* idx1,
                               const float* dist2, const int32_t* idx2, const float* grad_loss_b, float w1, float w2,
                               const float* cam, int H, int W, const void* records, const void* workspace,
                               const float* grad_total, float* grad_params, void* stream) {
    if (!cam || !records || !workspace || H <= 0 || W <= 0) return VPN_E_BADARG;
    RasterFinish rf;
    rf.rec = (const float4*)records;
    rf.ntile = ((W + R_TW - 1) / R_TW) * ((H + R_TH - 1) / R_TH);
    rf.words = (K + 63) / 64;
    rf.masks = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(records) + (size_t)B * K * R_REC * sizeof(float4));
    rf.partial = (const float*)workspace;
    rf.scale = grad_total;
    return launch_scb(params, kinds, u, seed, seed_dev, sample_base, B, K, n, points, gt_points, M, dist1, idx1, dist2, idx2,
                      grad_loss_b, w1, w2, grad_params, rf, stream);
}

// Backward of the WHOLE training step of the reference (train.py:243-264) in one launch: vpn_hotpath_bwd plus the EMD
// term, the VP-diversity term and (when its weight is not zero) the object-centred Chamfer term; see include/vpn_hip.h
extern "C" int vpn_trainstep_bwd(const float* params, const int32_t* kinds, uint64_t seed, const uint64_t* seed_dev,
                                 uint64_t sample_base, int B, int K, int n, const float* points, const float* gt_points, int M,
                                 const float* dist1, const int32_t* idx1, const float* dist2, const int32_t* idx2,
                                 float w1, float w2, const float* cam, int H, int W, const void* records,
                                 const void* workspace, const float* grad_total,
                                 const float* emd_dist, const int32_t* emd_assign, float emd_coef,
                                 const float* dv_dist1, const int32_t* dv_idx1, const float* dv_dist2, const int32_t* dv_idx2,
                                 float dv_c1, float dv_c2,
                                 const float* cn_points, const float* cn_gt, const float* cn_mat, const float* cn_dist1,
                                 const int32_t* cn_idx1, const float* cn_dist2, const int32_t* cn_idx2, float cn_c1, float cn_c2,
                                 int cn_M, float* grad_params, void* stream) {
    if (!grad_total) return VPN_E_BADARG;
    if (emd_dist && (!emd_assign || K * n != M)) return VPN_E_BADARG;              // emd_module.py:36: equal sizes
    if (dv_dist1 && (!dv_idx1 || !dv_dist2 || !dv_idx2)) return VPN_E_BADARG;
    if (cn_points && (!cn_gt || !cn_mat || !cn_dist1 || !cn_idx1 || !cn_dist2 || !cn_idx2 || cn_M <= 0)) return VPN_E_BADARG;
    RasterFinish rf;
    rf.scale = grad_total;
    if (records && workspace) {                                                   // the silhouette term's partials exist
        if (!cam || H <= 0 || W <= 0) return VPN_E_BADARG;
        rf.rec = (const float4*)records;
        rf.ntile = ((W + R_TW - 1) / R_TW) * ((H + R_TH - 1) / R_TH);
        rf.words = (K + 63) / 64;
        rf.masks = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(records) + (size_t)B * K * R_REC * sizeof(float4));
        rf.partial = (const float*)workspace;
    }
    StepExtras ex;
    ex.emd_dist = emd_dist; ex.emd_assign = emd_assign; ex.emd_coef = emd_coef;
    ex.dv_dist1 = dv_dist1; ex.dv_idx1 = dv_idx1; ex.dv_dist2 = dv_dist2; ex.dv_idx2 = dv_idx2; ex.dv_c1 = dv_c1; ex.dv_c2 = dv_c2;
    ex.cn_points = cn_points; ex.cn_gt = cn_gt; ex.cn_mat = cn_mat; ex.cn_dist1 = cn_dist1; ex.cn_idx1 = cn_idx1;
    ex.cn_dist2 = cn_dist2; ex.cn_idx2 = cn_idx2; ex.cn_c1 = cn_c1; ex.cn_c2 = cn_c2; ex.cn_M = cn_M;
    return launch_scb(params, kinds, nullptr, seed, seed_dev, sample_base, B, K, n, points, gt_points, M, dist1, idx1, dist2, idx2,
                      nullptr, w1, w2, grad_params, rf, stream, &ex);
}

static int mesh_check(const void* params, const void* kinds, const void* offsets, const void* ts, const void* tc,
                      int B, int K, int Ptot) {
    if (!params || !kinds || !offsets || (!ts && !tc)) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || Ptot <= 0) return VPN_E_BADARG;
    if (B > 65535 || (long long)B * Ptot * 3 > 0x7fffffffLL) return VPN_E_TOOBIG;
    return 0;
}

extern "C" int vpn_mesh_fwd(const float* params, const int32_t* kinds, const int32_t* offsets,
                            const float* tpl_sphere, const float* tpl_cuboid, int B, int K, int Ptot, float* verts,
                            void* stream) {
    int rc = mesh_check(params, kinds, offsets, tpl_sphere, tpl_cuboid, B, K, Ptot);
    if (rc) return rc;
    if (!verts) return VPN_E_BADARG;
    VPN_LAUNCH(mesh_fwd_kernel, dim3(K, B), dim3(MESH_BLOCK), 0, (hipStream_t)stream, params, kinds, offsets, tpl_sphere,
               tpl_cuboid, K, Ptot, verts);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_mesh_bwd(const float* params, const int32_t* kinds, const int32_t* offsets,
                            const float* tpl_sphere, const float* tpl_cuboid, int B, int K, int Ptot,
                            const float* grad_verts, float* grad_params, void* stream) {
    int rc = mesh_check(params, kinds, offsets, tpl_sphere, tpl_cuboid, B, K, Ptot);
    if (rc) return rc;
    if (!grad_verts || !grad_params) return VPN_E_BADARG;
    VPN_LAUNCH(mesh_bwd_kernel, dim3(K, B), dim3(MESH_BLOCK), 0, (hipStream_t)stream, params, kinds, offsets, tpl_sphere,
               tpl_cuboid, K, Ptot, grad_verts, grad_params);
    VPN_LAUNCH_CHECK();
    return 0;
}
