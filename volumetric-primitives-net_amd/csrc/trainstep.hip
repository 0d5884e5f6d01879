// trainstep.hip — the small kernels that complete the reference's training step (train.py:243-262) around the hot
// path: the VP-diversity loss's nearest neighbours (modules/loss/vp_diverse.py:12-18) and the reduction of every loss
// term into the weighted total (train.py:260).  gfx950.
#include "vpn_common.h"

#pragma clang fp contract(off)

namespace vpn {

constexpr int VD_THREADS = 256;
constexpr int VD_SLICES = 8;             // workgroups per sample: each takes a slice of the ground-truth cloud

// d2 -> the 64-bit key (IEEE sqrt bits, index): orders like "smaller distance, then lower index" (chamfer_distance.py:19-23)
__device__ inline unsigned long long vd_key(float d2, int e) {
    return ((unsigned long long)(unsigned)__float_as_int(sqrtf(d2)) << 32) | (unsigned)e;
}

// VPDiverseLoss.forward (vp_diverse.py:15-17): ChamferDistanceLoss(centres [B,K,3], gt [B,M,3], w1 = 0.5, w2 = 1.0), the
// centres being the K translations, read straight out of the packed parameters (the reference torch.cat's K views).
// Same arithmetic and tie rule as the Chamfer kernels: d = sqrt((dx^2 + dy^2) + dz^2), every operation rounded by itself,
// lowest index among equal DISTANCES.  The scans compare squared distances (no square root per pair); a pair whose
// squared distance is within 2 ulp of the running minimum -- the only ones whose root can tie with it -- goes through
// the exact key.  Grid (VD_SLICES, B): workgroup (s, b) does both directions for slice s of the ground truth; the centres'
// partial results keys_part [B][VD_SLICES][K] are merged by vpdiv_merge_kernel.
__global__ __launch_bounds__(VD_THREADS) void vpdiv_fwd_kernel(const float* __restrict__ params, const float* __restrict__ gt,
                                                               int K, int M, unsigned long long* __restrict__ keys_part,
                                                               float* __restrict__ dist2, int32_t* __restrict__ idx2) {
    extern __shared__ __attribute__((aligned(16))) float vd_lds[];          // K centres (x, y, z), nsl x K 64-bit keys, the slice's points
    float* cx = vd_lds; float* cy = cx + K; float* cz = cy + K;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(cz + K + (K & 1));
    const int nslk = max(1, VD_THREADS / K) * min(K, VD_THREADS);
    float* gs = reinterpret_cast<float*>(keys + nslk);                     // [3 * per]: the slice of the cloud (both scans read it 64+ times)
    const int sl = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* prm = params + (size_t)b * K * VPN_PARAM_STRIDE;
    for (int k = tid; k < K; k += VD_THREADS) { cx[k] = prm[k * VPN_PARAM_STRIDE + 7]; cy[k] = prm[k * VPN_PARAM_STRIDE + 8]; cz[k] = prm[k * VPN_PARAM_STRIDE + 9]; }
    __syncthreads();
    const float* Gg = gt + (size_t)b * M * 3;
    const int per = (M + VD_SLICES - 1) / VD_SLICES, e0 = sl * per, e1 = min(M, e0 + per);
    for (int i = tid; i < 3 * (e1 - e0); i += VD_THREADS) gs[i] = Gg[(size_t)e0 * 3 + i];
    __syncthreads();
    const float* G = gs - (size_t)e0 * 3;                                   // indexed by the global point number like the cloud itself
    auto near = [](float d2, float m) { return __float_as_int(d2) <= __float_as_int(m) + 2; };      // d2, m >= 0: bits order like values
    // direction 2: every ground-truth point's nearest centre
    for (int e = e0 + tid; e < e1; e += VD_THREADS) {
        const float gx = G[e * 3], gy = G[e * 3 + 1], gz = G[e * 3 + 2];
        float m = __builtin_inff(); unsigned long long key = ~0ull;
        for (int k = 0; k < K; ++k) {
            const float dx = cx[k] - gx, dy = cy[k] - gy, dz = cz[k] - gz;
            const float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
            if (d2 < m || near(d2, m)) { const unsigned long long c = vd_key(d2, k); key = c < key ? c : key; m = fminf(m, d2); }
        }
        dist2[(size_t)b * M + e] = __int_as_float((int)(unsigned)(key >> 32)); idx2[(size_t)b * M + e] = (int)(unsigned)key;
    }
    // direction 1: every centre's nearest ground-truth point of this slice: thread (sub-slice, centre)
    const int nsl = max(1, VD_THREADS / K);
    for (int k0 = 0; k0 < K; k0 += VD_THREADS) {
        const int k = k0 + (K >= VD_THREADS ? tid : tid % K), sidx = K >= VD_THREADS ? 0 : tid / K;
        if (k < K && sidx < nsl) {
            const int per2 = (e1 - e0 + nsl - 1) / nsl, f0 = e0 + sidx * per2, f1 = min(e1, f0 + per2);
            const float x = cx[k], y = cy[k], z = cz[k];
            float m = __builtin_inff(); unsigned long long key = ~0ull;
            for (int e = f0; e < f1; ++e) {
                const float dx = x - G[e * 3], dy = y - G[e * 3 + 1], dz = z - G[e * 3 + 2];
                const float d2 = ((dx * dx) + (dy * dy)) + (dz * dz);
                if (d2 < m || near(d2, m)) { const unsigned long long c = vd_key(d2, e); key = c < key ? c : key; m = fminf(m, d2); }
            }
            keys[sidx * K + (k - k0)] = key;
        }
        __syncthreads();
        for (int kk = tid; kk < min(K - k0, VD_THREADS); kk += VD_THREADS) {
            unsigned long long key = keys[kk];
            if (K < VD_THREADS) for (int s2 = 1; s2 < nsl; ++s2) { const unsigned long long c = keys[s2 * K + kk]; key = c < key ? c : key; }
            keys_part[((size_t)b * VD_SLICES + sl) * K + k0 + kk] = key;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void vpdiv_merge_kernel(const unsigned long long* __restrict__ keys_part, int BK, int K,
                                                          float* __restrict__ dist1, int32_t* __restrict__ idx1) {
    const int bk = blockIdx.x * 256 + threadIdx.x;
    if (bk >= BK) return;
    const int b = bk / K, k = bk - b * K;
    unsigned long long key = ~0ull;
    for (int s = 0; s < VD_SLICES; ++s) { const unsigned long long c = keys_part[((size_t)b * VD_SLICES + s) * K + k]; key = c < key ? c : key; }
    dist1[bk] = __int_as_float((int)(unsigned)(key >> 32)); idx1[bk] = (int)(unsigned)key;
}

// sum of f(v[i]) over i < count in a fixed order: thread-strided partial sums, lanes by butterfly, waves in order
template <bool SQRT, int THREADS>
__device__ inline float vd_block_sum(const float* __restrict__ v, int count, float* red) {
    float s = 0.0f;
    if (v) for (int i = threadIdx.x; i < count; i += THREADS) s += SQRT ? sqrtf(v[i]) : v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __syncthreads();                                                        // the previous sum's readers are done
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < THREADS / 64; ++w) t += red[w];
    return t;
}

// per sample: the sums the step's loss terms are means of -> part [B][8] = (sum sqrt(emd dist), sum cn_dist1, sum cn_dist2,
// sum dv_dist1, sum dv_dist2, 0, 0, 0)
constexpr int TP_THREADS = 256;
__global__ __launch_bounds__(TP_THREADS) void trainstep_partial_kernel(const float* __restrict__ emd_dist, const float* __restrict__ cn_dist1,
                                                                       const float* __restrict__ cn_dist2, float* __restrict__ dv_dist1,
                                                                       const float* __restrict__ dv_dist2, int N, int M, int Mc, int K,
                                                                       float* __restrict__ part,
                                                                       const unsigned long long* __restrict__ dv_keys, int32_t* __restrict__ dv_idx1) {
    const int b = blockIdx.x;
    if (dv_keys) {                                       // the centres' partial nearest neighbours of vpdiv_fwd_kernel: merged here
        for (int k = threadIdx.x; k < K; k += TP_THREADS) {
            unsigned long long key = ~0ull;
            for (int s = 0; s < VD_SLICES; ++s) { const unsigned long long c = dv_keys[((size_t)b * VD_SLICES + s) * K + k]; key = c < key ? c : key; }
            dv_dist1[(size_t)b * K + k] = __int_as_float((int)(unsigned)(key >> 32)); dv_idx1[(size_t)b * K + k] = (int)(unsigned)key;
        }
        __syncthreads();                                 // this workgroup sums what it has just written
    }
    // the five sums in ONE pass: every thread's loads are in flight together, one exchange through LDS (fixed order:
    // thread-strided partial sums, lanes by butterfly, waves in order)
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (emd_dist) for (int i = threadIdx.x; i < N; i += TP_THREADS) v[0] += sqrtf(emd_dist[(size_t)b * N + i]);     // torch.sqrt(dist), train.py:195
    if (cn_dist1) for (int i = threadIdx.x; i < N; i += TP_THREADS) v[1] += cn_dist1[(size_t)b * N + i];
    if (cn_dist2) for (int i = threadIdx.x; i < Mc; i += TP_THREADS) v[2] += cn_dist2[(size_t)b * Mc + i];
    if (dv_dist1) for (int i = threadIdx.x; i < K; i += TP_THREADS) v[3] += dv_dist1[(size_t)b * K + i];
    if (dv_dist2) for (int i = threadIdx.x; i < M; i += TP_THREADS) v[4] += dv_dist2[(size_t)b * M + i];
    __shared__ float red5[TP_THREADS / 64][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
        if ((threadIdx.x & 63) == 0) red5[threadIdx.x >> 6][q] = v[q];
    }
    __syncthreads();
    float t5[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { t5[q] = 0.0f; for (int w = 0; w < TP_THREADS / 64; ++w) t5[q] += red5[w][q]; }
    const float s_emd = t5[0], s_c1 = t5[1], s_c2 = t5[2], s_d1 = t5[3], s_d2 = t5[4];
    if (threadIdx.x == 0) {
        float* o = part + (size_t)b * 8;
        o[0] = s_emd; o[1] = s_c1; o[2] = s_c2; o[3] = s_d1; o[4] = s_d2; o[5] = 0.0f; o[6] = 0.0f; o[7] = 0.0f;
    }
}

// train.py:245-260: the five weighted terms and their sum.  hot [4] = (silhouette loss, depth loss, w_view * cd + w_sil * sil,
// mean_b cd_b) as vpn_raster_total_fwd_fin leaves them; part as above (samples added in order).
__global__ __launch_bounds__(64) void trainstep_finalize_kernel(const float* __restrict__ hot, const float* __restrict__ part, int has_emd,
                                                               int has_cn, int has_dv, int B, int N, int M, int Mc, int K,
                                                               float w_view, float w_can, float w_sil, float w_div, float w_emd,
                                                               float cd_w1, float cd_w2, float* __restrict__ out) {
    const int lane = threadIdx.x;
    // lane l adds the samples l, l + 64, ... of every column, then a butterfly over the lanes: a fixed order, all loads in flight
    float c[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = lane; b < B; b += 64) {
        const float4 p = *reinterpret_cast<const float4*>(part + (size_t)b * 8);
        c[0] += p.x; c[1] += p.y; c[2] += p.z; c[3] += p.w; c[4] += part[(size_t)b * 8 + 4];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c[i] += __shfl_xor(c[i], o, 64);
    const float s_emd = c[0], s_c1 = c[1], s_c2 = c[2], s_d1 = c[3], s_d2 = c[4];
    if (lane == 0) {
        const float view_cd = w_view * hot[3];
        const float obj_cd = has_cn ? w_can * (cd_w1 * (s_c1 / ((float)B * (float)N)) + cd_w2 * (s_c2 / ((float)B * (float)Mc))) : 0.0f;
        const float sil = w_sil * hot[0];
        const float div = has_dv ? w_div * (0.5f * (s_d1 / ((float)B * (float)K)) + 1.0f * (s_d2 / ((float)B * (float)M))) : 0.0f;
        const float emd = has_emd ? w_emd * (s_emd / ((float)B * (float)N)) : 0.0f;
        out[0] = view_cd; out[1] = obj_cd; out[2] = sil; out[3] = div; out[4] = emd;
        out[5] = (((view_cd + obj_cd) + sil) + div) + emd;                             // train.py:260
    }
}

}  // namespace vpn

using namespace vpn;

extern "C" size_t vpn_vpdiv_workspace(int B, int K) { return B > 0 && K > 0 ? (size_t)B * VD_SLICES * K * sizeof(unsigned long long) : 0; }

extern "C" int vpn_vpdiv_fwd(const float* params, const float* gt_points, int B, int K, int M, float* dist1, int32_t* idx1,
                             float* dist2, int32_t* idx2, void* workspace, void* stream) {
    if (!params || !gt_points || !dist2 || !idx2 || !workspace || B <= 0 || K <= 0 || M <= 0) return VPN_E_BADARG;
    if ((dist1 != nullptr) != (idx1 != nullptr) || ((uintptr_t)workspace & 7) != 0) return VPN_E_BADARG;
    if (K > VPN_MAX_PRIMS || B > 65535 || B > 0x7fffffff / max(K, M)) return VPN_E_TOOBIG;
    const int nsl = max(1, VD_THREADS / K);
    const int per = (M + VD_SLICES - 1) / VD_SLICES;
    const size_t lds = (size_t)(3 * K + (K & 1)) * sizeof(float) + (size_t)nsl * min(K, VD_THREADS) * sizeof(unsigned long long)
                       + (size_t)3 * per * sizeof(float);
    if (lds > 64 * 1024) return VPN_E_TOOBIG;
    unsigned long long* keys_part = (unsigned long long*)workspace;
    VPN_LAUNCH(vpdiv_fwd_kernel, dim3(VD_SLICES, B), dim3(VD_THREADS), lds, (hipStream_t)stream, params, gt_points, K, M, keys_part, dist2, idx2);
    VPN_LAUNCH_CHECK();
    if (!dist1) return 0;                                // the caller merges (vpn_trainstep_finalize does, in its per-sample pass)
    VPN_LAUNCH(vpdiv_merge_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, keys_part, B * K, K, dist1, idx1);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t vpn_trainstep_workspace(int B) { return B > 0 ? (size_t)B * 8 * sizeof(float) : 0; }

extern "C" int vpn_trainstep_finalize(const float* hot_losses, const float* emd_dist, const float* cn_dist1, const float* cn_dist2,
                                      const float* dv_dist1, const float* dv_dist2, int B, int N, int M, int Mc, int K,
                                      float w_view, float w_can, float w_sil, float w_div, float w_emd, float cd_w1, float cd_w2,
                                      void* workspace, const void* dv_workspace, float* dv_dist1_out, int32_t* dv_idx1_out,
                                      float* out, void* stream) {
    if (!hot_losses || !out || !workspace || B <= 0 || N <= 0 || M <= 0 || K <= 0) return VPN_E_BADARG;
    if (dv_workspace && (!dv_dist1_out || !dv_idx1_out || !dv_dist2 || dv_dist1)) return VPN_E_BADARG;
    if ((cn_dist1 != nullptr) != (cn_dist2 != nullptr) || ((dv_dist1 != nullptr || dv_workspace != nullptr) != (dv_dist2 != nullptr))) return VPN_E_BADARG;
    if (cn_dist1 && Mc <= 0) return VPN_E_BADARG;
    float* part = (float*)workspace;
    // dv_workspace given: the VP-diversity keys of vpn_vpdiv_fwd(dist1 = NULL) are merged into dv_dist1_out / dv_idx1_out first
    float* d1 = dv_workspace ? dv_dist1_out : const_cast<float*>(dv_dist1);
    VPN_LAUNCH(trainstep_partial_kernel, dim3(B), dim3(TP_THREADS), 0, (hipStream_t)stream, emd_dist, cn_dist1, cn_dist2, d1, dv_dist2,
               N, M, Mc, K, part, (const unsigned long long*)dv_workspace, dv_idx1_out);
    VPN_LAUNCH_CHECK();
    VPN_LAUNCH(trainstep_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hot_losses, part, emd_dist ? 1 : 0, cn_dist1 ? 1 : 0,
               (dv_dist1 || dv_workspace) ? 1 : 0, B, N, M, Mc, K, w_view, w_can, w_sil, w_div, w_emd, cd_w1, cd_w2, out);
    VPN_LAUNCH_CHECK();
    return 0;
}
