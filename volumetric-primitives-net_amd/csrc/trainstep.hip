// trainstep.hip — the small kernels that complete the reference's training step (train.py:243-262) around the hot
// path: the VP-diversity loss's nearest neighbours (modules/loss/vp_diverse.py:12-18) and the reduction of every loss
// term into the weighted total (train.py:260).  gfx950.
#include "vpn_common.h"

#pragma clang fp contract(off)

namespace vpn {

constexpr int VD_THREADS = 1024;

// VPDiverseLoss.forward (vp_diverse.py:15-17): ChamferDistanceLoss(centres [B,K,3], gt [B,M,3], w1 = 0.5, w2 = 1.0), the
// centres being the K translations.  One workgroup per sample: the centres are read straight out of the packed
// parameters (the reference torch.cat's K views), both directions in one pass over the ground truth.  Same arithmetic
// and tie rule as the Chamfer kernels: d = sqrt((dx^2 + dy^2) + dz^2), every operation rounded by itself, lowest index
// among equal distances (chamfer_distance.py:14-23) -- the 64-bit key (distance bits, index) orders exactly like that.
__global__ __launch_bounds__(VD_THREADS) void vpdiv_fwd_kernel(const float* __restrict__ params, const float* __restrict__ gt,
                                                               int K, int M, float* __restrict__ dist1, int32_t* __restrict__ idx1,
                                                               float* __restrict__ dist2, int32_t* __restrict__ idx2) {
    extern __shared__ __attribute__((aligned(16))) float vd_lds[];          // K centres (x, y, z), then nsl x K 64-bit keys
    float* cx = vd_lds; float* cy = cx + K; float* cz = cy + K;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(cz + K + (K & 1));
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* prm = params + (size_t)b * K * VPN_PARAM_STRIDE;
    for (int k = tid; k < K; k += VD_THREADS) { cx[k] = prm[k * VPN_PARAM_STRIDE + 7]; cy[k] = prm[k * VPN_PARAM_STRIDE + 8]; cz[k] = prm[k * VPN_PARAM_STRIDE + 9]; }
    __syncthreads();
    const float* G = gt + (size_t)b * M * 3;
    // direction 2: every ground-truth point's nearest centre
    for (int e = tid; e < M; e += VD_THREADS) {
        const float gx = G[e * 3], gy = G[e * 3 + 1], gz = G[e * 3 + 2];
        float best = __builtin_inff(); int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float dx = cx[k] - gx, dy = cy[k] - gy, dz = cz[k] - gz;
            const float d = sqrtf(((dx * dx) + (dy * dy)) + (dz * dz));
            if (d < best) { best = d; bi = k; }                             // strict: the lowest index among equals stays
        }
        dist2[(size_t)b * M + e] = best; idx2[(size_t)b * M + e] = bi;
    }
    // direction 1: every centre's nearest ground-truth point: thread (slice s, centre k) scans its slice of the cloud
    const int nsl = max(1, VD_THREADS / K);
    for (int k0 = 0; k0 < K; k0 += VD_THREADS) {                            // K <= 1024: one round
        const int k = k0 + tid % max(K, 1), sidx = tid / max(K, 1);
        if (K <= VD_THREADS && sidx < nsl && k < K) {
            const int per = (M + nsl - 1) / nsl, e0 = sidx * per, e1 = min(M, e0 + per);
            const float x = cx[k], y = cy[k], z = cz[k];
            unsigned long long key = ~0ull;
            for (int e = e0; e < e1; ++e) {
                const float dx = x - G[e * 3], dy = y - G[e * 3 + 1], dz = z - G[e * 3 + 2];
                const float d = sqrtf(((dx * dx) + (dy * dy)) + (dz * dz));
                const unsigned long long c = ((unsigned long long)(unsigned)__float_as_int(d) << 32) | (unsigned)e;
                key = c < key ? c : key;
            }
            keys[sidx * K + k] = key;
        }
    }
    __syncthreads();
    for (int k = tid; k < K; k += VD_THREADS) {
        unsigned long long key = keys[k];
        for (int sl = 1; sl < nsl; ++sl) { const unsigned long long c = keys[sl * K + k]; key = c < key ? c : key; }
        dist1[b * K + k] = __int_as_float((int)(unsigned)(key >> 32)); idx1[b * K + k] = (int)(unsigned)key;
    }
}

// sum of f(v[i]) over i < count in a fixed order: thread-strided partial sums, lanes by butterfly, waves in order
template <bool SQRT>
__device__ inline float vd_block_sum(const float* __restrict__ v, long long count, float* red) {
    float s = 0.0f;
    if (v) for (long long i = threadIdx.x; i < count; i += VD_THREADS) s += SQRT ? sqrtf(v[i]) : v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __syncthreads();                                                        // the previous sum's readers are done
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < VD_THREADS / 64; ++w) t += red[w];
    return t;
}

// train.py:245-260: the five weighted terms and their sum.  hot [4] = (silhouette loss, depth loss, w_view * cd + w_sil * sil,
// mean_b cd_b) as vpn_raster_total_fwd_fin leaves them.
__global__ __launch_bounds__(VD_THREADS) void trainstep_finalize_kernel(const float* __restrict__ hot, const float* __restrict__ emd_dist,
                                                                        const float* __restrict__ cn_dist1, const float* __restrict__ cn_dist2,
                                                                        const float* __restrict__ dv_dist1, const float* __restrict__ dv_dist2,
                                                                        int B, int N, int M, int Mc, int K, float w_view, float w_can,
                                                                        float w_sil, float w_div, float w_emd, float cd_w1, float cd_w2,
                                                                        float* __restrict__ out) {
    __shared__ float red[VD_THREADS / 64];
    const float s_emd = vd_block_sum<true>(emd_dist, (long long)B * N, red);          // torch.sqrt(dist).mean(), train.py:195
    const float s_c1 = vd_block_sum<false>(cn_dist1, (long long)B * N, red), s_c2 = vd_block_sum<false>(cn_dist2, (long long)B * Mc, red);
    const float s_d1 = vd_block_sum<false>(dv_dist1, (long long)B * K, red), s_d2 = vd_block_sum<false>(dv_dist2, (long long)B * M, red);
    if (threadIdx.x == 0) {
        const float view_cd = w_view * hot[3];
        const float obj_cd = cn_dist1 ? w_can * (cd_w1 * (s_c1 / ((float)B * (float)N)) + cd_w2 * (s_c2 / ((float)B * (float)Mc))) : 0.0f;
        const float sil = w_sil * hot[0];
        const float div = dv_dist1 ? w_div * (0.5f * (s_d1 / ((float)B * (float)K)) + 1.0f * (s_d2 / ((float)B * (float)M))) : 0.0f;
        const float emd = emd_dist ? w_emd * (s_emd / ((float)B * (float)N)) : 0.0f;
        out[0] = view_cd; out[1] = obj_cd; out[2] = sil; out[3] = div; out[4] = emd;
        out[5] = (((view_cd + obj_cd) + sil) + div) + emd;                             // train.py:260
    }
}

}  // namespace vpn

using namespace vpn;

extern "C" int vpn_vpdiv_fwd(const float* params, const float* gt_points, int B, int K, int M, float* dist1, int32_t* idx1,
                             float* dist2, int32_t* idx2, void* stream) {
    if (!params || !gt_points || !dist1 || !idx1 || !dist2 || !idx2 || B <= 0 || K <= 0 || M <= 0) return VPN_E_BADARG;
    if (K > VD_THREADS || B > 0x7fffffff / max(K, M)) return VPN_E_TOOBIG;
    const int nsl = max(1, VD_THREADS / K);
    const size_t lds = (size_t)(3 * K + (K & 1)) * sizeof(float) + (size_t)nsl * K * sizeof(unsigned long long);
    VPN_LAUNCH(vpdiv_fwd_kernel, dim3(B), dim3(VD_THREADS), lds, (hipStream_t)stream, params, gt_points, K, M, dist1, idx1, dist2, idx2);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_trainstep_finalize(const float* hot_losses, const float* emd_dist, const float* cn_dist1, const float* cn_dist2,
                                      const float* dv_dist1, const float* dv_dist2, int B, int N, int M, int Mc, int K,
                                      float w_view, float w_can, float w_sil, float w_div, float w_emd, float cd_w1, float cd_w2,
                                      float* out, void* stream) {
    if (!hot_losses || !out || B <= 0 || N <= 0 || M <= 0 || K <= 0) return VPN_E_BADARG;
    if ((cn_dist1 != nullptr) != (cn_dist2 != nullptr) || (dv_dist1 != nullptr) != (dv_dist2 != nullptr)) return VPN_E_BADARG;
    if (cn_dist1 && Mc <= 0) return VPN_E_BADARG;
    VPN_LAUNCH(trainstep_finalize_kernel, dim3(1), dim3(VD_THREADS), 0, (hipStream_t)stream, hot_losses, emd_dist, cn_dist1, cn_dist2,
               dv_dist1, dv_dist2, B, N, M, Mc, K, w_view, w_can, w_sil, w_div, w_emd, cd_w1, cd_w2, out);
    VPN_LAUNCH_CHECK();
    return 0;
}
