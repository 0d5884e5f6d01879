// raster.hip — differentiable soft raster of volumetric primitives (ellipsoids and
// cuboids) straight from the packed [B,K,10] parameter tensor: silhouette + soft-min
// depth, forward and analytic backward, for gfx950.
//
// Sits behind VertexRenderer.render (modules/render/vertex_renderer.py:14-26) and the
// per-sample loop of SilhouetteLoss.forward (modules/loss/silhouette.py:16-20).  The
// reference meshes each primitive (modules/meshing/sphere.py:8-27) and hands the mesh
// to kaolin's DIBRenderer, which is not part of its tree: this operator is new and its
// specification is oracle/vpn_oracle.py::raster (parity unpinned w.r.t. kaolin; geometry
// pinned to reference-derived data by tests/test_raster_geometry.py).
//
// Work decomposition (CDNA4):
//   * records (raster_prep_kernel in sampler.hip, or the sampler's own forward launch in the hot-path step): per
//     (image, primitive) the pose from q, the camera-space ray coefficients (o~, Mr, Mu, Mf), the exact culling
//     conic and its pixel box, then the camera basis and the pose for the backward chain rule -> R_REC float4
//     [B,K,14] (vpn_raster_common.h);
//   * tile kernels: ONE WAVEFRONT = ONE WORKGROUP PER 16x16 PIXEL TILE, 1-D grid in centre-out tile order with the
//     images interleaved (the heavy tiles start first, every XCD gets the same mix).  The wave first builds the
//     tile's primitive mask itself (lane = primitive: pixel box, then exact conic-vs-rectangle; a primitive is
//     dropped only where its coverage logit is below -X_CUT: coverage < 1.2e-7), stages the records of the
//     visible primitives in its own LDS slice and loops over them with scalar control flow.  Lane l owns one
//     pixel in each 8x8 quadrant of the tile; a primitive is evaluated per quadrant only where it can reach it;
//   * raster_total_kernel: forward AND backward of the training step's image losses in one pass (state in
//     registers, no `aux`, GT read once); raster_fwd / raster_bwd kernels: the general two-call form (images or
//     two separately weighted losses), with the per-pixel state saved in `aux`;
//   * backward: the per-pixel gradients w.r.t. the 12 ray coefficients of a primitive are summed
//     over the lane's pixels, reduced over the wave with a transposing butterfly (17 shuffles for
//     12 values) and written as 48 contiguous bytes of partial[b][k][tile] for the pairs of the mask only;
//     a finishing kernel sums them in a fixed order and applies the chain rule to (v,q,t).  No atomics in the
//     gradient: bitwise reproducible;
//   * loss_finalize_kernel: per-sample sums of the tile losses (and of the Chamfer minima) and the batch
//     totals by the last-arriving workgroup, in a fixed order.
#include "vpn_raster_common.h"
#include <type_traits>

#define R_EXP(x) __expf(x)   // v_exp_f32 based; the raster is a 1e-4 contract
// a / b as a * v_rcp_f32(b): the compiler's fp32 division is a ~8-instruction sequence (range scaling for huge and
// denormal divisors) even with the 1-ulp option, and these kernels divide 3-6 times per pixel x primitive
#define R_RCP(x) __builtin_amdgcn_rcpf(x)

namespace vpn {

// per pixel x primitive forward quantities
struct PixPrim {
    float d[3];
    float m2, z;
    float a, c, E, wgt;
    bool xin, ein;
    // ellipsoid intermediates
    float A, invA, Bq, s, wv[3], rr, h, chord;
    // cuboid intermediates
    float lam, n, den, L, tn, dsafe;
    int sel, zi;
};

template <int KIND>   // compile-time kind: the caller branches once per primitive (wave-uniform), not once per pixel
__device__ inline void eval_prim(const float4 r0, const float4 r1, const float4 r2, const float4 r3, float px,
                                 float py, float inv_sigma, float inv_gamma, float zref, PixPrim& q) {
    const float o[3] = {r0.x, r0.y, r0.z};
    constexpr int kind = KIND;
    q.d[0] = r3.x + px * r1.x + py * r2.x;
    q.d[1] = r3.y + px * r1.y + py * r2.y;
    q.d[2] = r3.z + px * r1.z + py * r2.z;
    if constexpr (kind == VPN_SPHERE) {
        q.A = q.d[0] * q.d[0] + q.d[1] * q.d[1] + q.d[2] * q.d[2];
        q.Bq = o[0] * q.d[0] + o[1] * q.d[1] + o[2] * q.d[2];
        q.invA = R_RCP(q.A);
        q.s = -q.Bq * q.invA;
        q.wv[0] = o[0] + q.s * q.d[0]; q.wv[1] = o[1] + q.s * q.d[1]; q.wv[2] = o[2] + q.s * q.d[2];
        q.m2 = q.wv[0] * q.wv[0] + q.wv[1] * q.wv[1] + q.wv[2] * q.wv[2];
        const float u = 1.0f - q.m2;                 // h = squareplus(u), cancellation-free for u < 0
        q.rr = sqrtf(u * u + R_EPS_H);
        q.h = u >= 0.0f ? 0.5f * (u + q.rr) : (0.5f * R_EPS_H) * R_RCP(q.rr - u);
        q.chord = sqrtf(q.h * q.invA);
        q.z = q.s - q.chord;
    } else {
        float ad[3] = {fabsf(q.d[0]), fabsf(q.d[1]), fabsf(q.d[2])};
        float n01 = o[1] * q.d[0] - o[0] * q.d[1], d01 = ad[0] + ad[1] + R_EPS_D;
        float n02 = o[2] * q.d[0] - o[0] * q.d[2], d02 = ad[0] + ad[2] + R_EPS_D;
        float n12 = o[2] * q.d[1] - o[1] * q.d[2], d12 = ad[1] + ad[2] + R_EPS_D;
        float l01 = fabsf(n01) * R_RCP(d01), l02 = fabsf(n02) * R_RCP(d02), l12 = fabsf(n12) * R_RCP(d12);
        q.lam = l01; q.n = n01; q.den = d01; q.sel = 0;
        if (l02 > q.lam) { q.lam = l02; q.n = n02; q.den = d02; q.sel = 1; }
        if (l12 > q.lam) { q.lam = l12; q.n = n12; q.den = d12; q.sel = 2; }
        q.L = fmaxf(q.lam, 1.0f);
        float tn[3], ds[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float sg = q.d[i] < 0.0f ? -1.0f : 1.0f;
            ds[i] = ad[i] < R_EPS_D ? sg * R_EPS_D : q.d[i];
            tn[i] = -(q.L * sg + o[i]) * R_RCP(ds[i]);
        }
        q.z = tn[0]; q.zi = 0;
        if (tn[1] > q.z) { q.z = tn[1]; q.zi = 1; }
        if (tn[2] > q.z) { q.z = tn[2]; q.zi = 2; }
        q.tn = q.z;
        q.dsafe = q.zi == 0 ? ds[0] : (q.zi == 1 ? ds[1] : ds[2]);
        q.m2 = q.lam * q.lam;
    }
    // coverage a = sigmoid(x), c = 1 - a = sigmoid(-x): with x clamped to +-80 exp(-x) stays finite in fp32, so one
    // exp and one rcp give both without a sign split (a = 1/(1+e), c = e a: no cancellation on either side).
    // Slow-class instructions (v_cmp / v_cndmask / v_min / v_max / v_med3: 1.7x a v_fma on gfx950) are
    // kept to a minimum here: one clamp and one compare per logit.
    const float xr = (1.0f - q.m2) * inv_sigma;
    q.xin = fabsf(xr) <= R_X_CLAMP;
    const float x = __builtin_amdgcn_fmed3f(xr, -R_X_CLAMP, R_X_CLAMP);
    const float en = R_EXP(-x);
    q.a = R_RCP(1.0f + en);
    q.c = en * q.a;
    const float er = (zref - q.z) * inv_gamma;
    q.ein = fabsf(er) <= R_E_CLAMP;
    const float e = __builtin_amdgcn_fmed3f(er, -R_E_CLAMP, R_E_CLAMP);
    q.E = R_EXP(e);
    q.wgt = q.a * q.E;
}

// gradient of the loss w.r.t. the primitive's ray coefficients through one pixel:
// go = dL/do~ (3), gd = dL/dd~ (3), given gz = dL/dz and gm2 = dL/dm2 of this pixel x primitive
template <int KIND>
__device__ inline void prim_backward(const float4 r0, const PixPrim& q, float gz, float gm2, float go[3],
                                     float gd[3]) {
    const float o[3] = {r0.x, r0.y, r0.z};
    if constexpr (KIND == VPN_SPHERE) {
        const float gchord = -gz;                       // z = s - chord, chord = sqrt(h * invA)
        // d chord/d u = (0.5/chord) invA dh/du with dh/du = h/r  ->  0.5 chord / r ;  u = 1 - m2
        gm2 -= gchord * 0.5f * q.chord * R_RCP(q.rr);
        float ginvA = gchord * 0.5f * q.chord * q.A;    // d chord/d invA = 0.5 h / chord = 0.5 chord A
        float gs = gz;
        const float gwv[3] = {2.0f * gm2 * q.wv[0], 2.0f * gm2 * q.wv[1], 2.0f * gm2 * q.wv[2]};   // m2 = w.w
        gs += gwv[0] * q.d[0] + gwv[1] * q.d[1] + gwv[2] * q.d[2];                                  // w = o + s d
        const float gBq = -gs * q.invA;                 // s = -Bq * invA
        ginvA -= gs * q.Bq;
        const float gAq = -ginvA * q.invA * q.invA;     // invA = 1 / (d.d)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            go[i] = gwv[i] + gBq * q.d[i];
            gd[i] = q.s * gwv[i] + 2.0f * gAq * q.d[i] + gBq * o[i];
        }
    } else {
        // z = tn[zi] = -(L sg + o_zi) / dsafe
        const int zi = q.zi;
        const float dzi = zi == 0 ? q.d[0] : (zi == 1 ? q.d[1] : q.d[2]);
        const float sg = dzi < 0.0f ? -1.0f : 1.0f;
        const float ids = R_RCP(q.dsafe);
        const float gL = -gz * sg * ids;
        const float go_z = -gz * ids;
        const float gd_z = fabsf(dzi) < R_EPS_D ? 0.0f : -gz * q.tn * ids;
        float glam = 2.0f * q.lam * gm2;                // m2 = lam^2
        if (q.lam >= 1.0f) glam += gL;                  // L = max(lam, 1)
        // lam = |n| / den
        const float sn = q.n > 0.0f ? 1.0f : (q.n < 0.0f ? -1.0f : 0.0f);
        const float iden = R_RCP(q.den);
        const float gn = glam * sn * iden;
        const float gden = -glam * q.lam * iden;
        // pair (i,j): sel 0 -> (0,1), 1 -> (0,2), 2 -> (1,2);  n = o_j d_i - o_i d_j, den = |d_i| + |d_j| + eps
        const int pi = q.sel == 2 ? 1 : 0, pj = q.sel == 0 ? 1 : 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float g_o = 0.f, g_d = 0.f;
            const float sd = q.d[i] > 0.0f ? 1.0f : (q.d[i] < 0.0f ? -1.0f : 0.0f);
            if (i == pi) {
                const float oj = pj == 1 ? o[1] : o[2], dj = pj == 1 ? q.d[1] : q.d[2];
                g_d += gn * oj + gden * sd;
                g_o -= gn * dj;
            }
            if (i == pj) {
                const float oi = pi == 0 ? o[0] : o[1], di = pi == 0 ? q.d[0] : q.d[1];
                g_o += gn * di;
                g_d += -gn * oi + gden * sd;
            }
            if (i == zi) { g_o += go_z; g_d += gd_z; }
            go[i] = g_o; gd[i] = g_d;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Visibility of (tile, primitive) pairs.  The FORWARD tile kernels test their tile against the primitives (one
// primitive per lane: pixel box first, then the exact conic-vs-rectangle test; a primitive is dropped only where its
// coverage logit is below -X_CUT on the whole tile) and store the result,
//   masks[(b * ntile + tile) * words + w] bit j  <=>  primitive 64 w + j may touch the 16x16 tile,
// which the backward and finishing kernels read back instead of repeating the test.

// (prim_hits_tile: vpn_raster_common.h -- shared with the launch that sorts the tiles of the training step)

// ---------------------------------------------------------------------------------------------------------------
// Tile kernels.  ONE WAVEFRONT (= one workgroup) PER 16x16 PIXEL TILE.  The visible primitives of the tile come from
// the tile mask the wave builds itself (wave-uniform: the loop is scalar control flow); their records are staged in
// the wave's LDS slice and read from there with vector loads (an SGPR operand makes a VALU instruction slow-class).

// Lane l owns one pixel in each 8x8 QUADRANT of the tile: slot s = quadrant (s & 1, s >> 1), position (l & 7, l >> 3)
// inside it.  A primitive is evaluated per quadrant only where it can reach it (second, 8x8 level of the cull).
struct Tile {
    int b, tile, c0, r0;
    int colq[2], rowq[2];       // this lane's column in quadrant column 0 / 1, row in quadrant row 0 / 1
    float pxq[2], pyq[2];
    __device__ int col(int s) const { return colq[s & 1]; }
    __device__ int row(int s) const { return rowq[s >> 1]; }
    __device__ float px(int s) const { return pxq[s & 1]; }
    __device__ float py(int s) const { return pyq[s >> 1]; }
};

// One wave (= one workgroup) per tile, 1-D grid of ntile * B workgroups: id = p * B + b with p running over the tiles
// CENTRE FIRST (rows and columns in the order n/2, n/2 - 1, n/2 + 1, ...: a bijection of [0, n) for every n).  The
// primitives crowd the image centre, a tile's cost is proportional to its primitive count (0 to ~10 at C3) and a
// heavy tile started late runs on alone at the end of the launch: heavy-first order shortens that tail; one-wave
// workgroups because a 4-wave workgroup holds its 4 slots until its slowest tile is done; id % 8 = b % 8 keeps every
// XCD on whole images with the same mix of tiles (speed only: any order is correct).
// `tile_of_entry` (training step): the tile of image b with rank r when its tiles are sorted by the number of
// visible primitives, heaviest first (written by the launch that wrote the tile masks).  A tile's cost is proportional
// to that number (0 .. 10 at C3, and it varies from image to image at one position), the launch lasts at least as long
// as its heaviest tile (~30 us), and with the position-based order a heavy tile away from the centre started in the
// second or third round of waves and WAS the tail.
__device__ inline Tile make_tile(int H, int W, int tiles_x, int tiles_y, int B, int tile_of_entry = -1) {
    Tile T;
    const int lane = threadIdx.x & 63;
    const int id = blockIdx.x;
    const int pt = id / B;
    T.b = id - pt * B;
    const int iy = pt / tiles_x, ix = pt - iy * tiles_x;
    auto co = [](int i, int n) { const int c = n >> 1, h = (i + 1) >> 1; return (i & 1) ? c - h : c + h; };
    int tx = co(ix, tiles_x), ty = co(iy, tiles_y);
    if (tile_of_entry >= 0) {
        const int t = min(tile_of_entry, tiles_x * tiles_y - 1);
        ty = t / tiles_x; tx = t - ty * tiles_x;
    }
    T.tile = ty * tiles_x + tx;
    T.c0 = tx * R_TW; T.r0 = ty * R_TH;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        T.colq[h] = T.c0 + 8 * h + (lane & 7);
        T.rowq[h] = T.r0 + 8 * h + (lane >> 3);
        T.pxq[h] = ((2.0f * ((float)T.colq[h] + 0.5f) / (float)W) - 1.0f) * (R_TAN_HALF_FOV * (float)W / (float)H);
        T.pyq[h] = (1.0f - (2.0f * ((float)T.rowq[h] + 0.5f) / (float)H)) * R_TAN_HALF_FOV;
    }
    return T;
}

// A VALU instruction with an SGPR source issues at 0.6x the rate of one with VGPR sources on gfx950 (1.77 vs 1.07 ns
// per wave-instruction, tools/ubench/valu_rates2.hip), and every use of a ray coefficient would be one; a vector load
// at the top of each primitive's iteration exposes its L1/L2 latency instead.  So each wave first copies the
// records of its tile's visible primitives into a private LDS array (lane i of a mask word fetches primitive i if
// its bit is set and stores it at slot = number of set bits below i), and the loops read them from LDS with
// wave-uniform addresses (broadcast) into VGPRs.  `vzero()` is a zero the compiler cannot see through (keeps
// uniform scalars of the pixel loops in VGPRs).
__device__ inline int vzero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

__device__ inline unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int R_SLOT = 4;            // float4 per staged primitive: (o~|kind, Mr|k, Mu, Mf)
// LDS of a tile wave (dynamic): `nslot` staged records, then their culling records.  nslot = min(K, 64): a mask word's
// worth -- 4.5 KB at K = 32, 9 KB from K = 64 on
__host__ __device__ inline int raster_nslot(int K) { return K < 64 ? K : 64; }
constexpr int R_RED_FLOATS = 12 * 64;      // transposing reduction of a tile's 12 gradient sums through LDS (reduce12_lds)
__host__ __device__ inline size_t raster_lds_bytes(int K) {
    return (size_t)raster_nslot(K) * (R_SLOT + R_CULL) * sizeof(float4) + 4 * sizeof(unsigned long long) + R_RED_FLOATS * sizeof(float);
}

// Mask word w of this tile and staging of its visible primitives in one go: lane i fetches the whole record of
// primitive 64 w + i (7 float4, one round trip), tests it against the tile, and if visible stores its ray coefficients
// at slot = number of visible primitives below i.  `mask_in`: a mask computed earlier (backward kernels) instead of the
// test; `mask_out`: where lane 0 stores the mask (forward kernels).  Returns the mask (wave-uniform).
// quadrant words of the tile that arrived with its entry (training step): they are put where quadrant_bits_all would
// have left them (LDS, behind the culling records) and the loops read them from there
// (`quad_ready` of the tile loops: a plain flag -- as a one-byte struct it was written to and re-read from scratch memory)

__device__ inline unsigned long long stage_word(const Tile& T, const float4* __restrict__ rec_b, int w, int K, int H, int W,
                                                float4* srec, const unsigned long long* mask_in,
                                                unsigned long long* mask_out, bool need_cull = true, bool has_m = false,
                                                unsigned long long m_val = 0ull) {
    const int lane = threadIdx.x & 63;
    const int k = w * 64 + lane;
    const float4* rk = rec_b + (size_t)(k < K ? k : 0) * R_REC;
    unsigned long long m;
    float4 a = rk[0], b = rk[1], c = rk[2], d = rk[3];
    if (has_m) {
        m = m_val;                                                   // came with the tile's entry (wave-uniform)
    } else if (mask_in) {
        m = uniform64(mask_in[w]);
    } else {
        float4 cr[R_CULL];
#pragma unroll
        for (int i = 0; i < R_CULL; ++i) cr[i] = rk[4 + i];
        m = __ballot(k < K && prim_hits_tile(__float_as_int(a.w), cr, T.c0, T.r0, H, W));
        if (mask_out && lane == 0) mask_out[w] = m;
    }
    if ((m >> lane) & 1ull) {
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        b.w = __int_as_float(k);
        srec[slot * R_SLOT + 0] = a; srec[slot * R_SLOT + 1] = b; srec[slot * R_SLOT + 2] = c; srec[slot * R_SLOT + 3] = d;
        float4* scull = srec + raster_nslot(K) * R_SLOT + slot * R_CULL;
        const bool box = __float_as_int(a.w) != VPN_SPHERE;                  // an ellipsoid's conic ends at float4 2
        if (need_cull) {
#pragma unroll
            for (int i = 0; i < R_CULL; ++i) if (i < 3 || box) scull[i] = rk[4 + i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    return m;
}

// Second level of the cull: which of the tile's four 8x8 quadrants can staged primitive j reach?  Lane l tests
// (slot l >> 2, quadrant l & 3) of 16 slots per pass; the ballot holds 4 bits per slot.  Returns the ballot of pass
// `pass` (slots 16 pass .. 16 pass + 15), wave-uniform.
__device__ inline unsigned long long quadrant_bits(const Tile& T, const float4* srec, int n, int pass, int K, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int slot = pass * 16 + (lane >> 2), qd = lane & 3;
    bool vis = false;
    if (slot < n) {
        const float4* scull = srec + raster_nslot(K) * R_SLOT + slot * R_CULL;
        vis = prim_hits_tile(__float_as_int(srec[slot * R_SLOT].w), scull, T.c0 + 8 * (qd & 1), T.r0 + 8 * (qd >> 1), H, W, 8, 8);
    }
    return __ballot(vis);
}

// 12 per-lane values -> lane L (L < 48) holds the wave total of value L >> 2, THROUGH LDS: every lane writes its 12 values
// value-major, lane 4 i + q reads back the 16 contributions of lanes 16 q .. 16 q + 15 to value i (four 16-byte reads),
// adds them in order, and two quad steps add the four quarters.  The tile kernels are bound by VALU issue and the LDS pipe
// issues beside it: this is 17 VALU instructions where the register butterfly (wave_reduce16) is 30 selects + 17 adds.
// Fixed order: bitwise reproducible (not the butterfly's order: the last bits differ from it).
__device__ inline float reduce12_lds(const float v[16], float* red) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 12; ++i) red[i * 64 + lane] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float r = 0.0f;
    if (lane < 48) {
        const float4* p = reinterpret_cast<const float4*>(red + (lane >> 2) * 64 + (lane & 3) * 16);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        r = ((((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w))));
    }
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    __builtin_amdgcn_wave_barrier();           // the next primitive's writes come after these reads
    return r;
}
__device__ inline float* reduce_scratch(float4* srec, int K) {
    return reinterpret_cast<float*>(srec + raster_nslot(K) * (R_SLOT + R_CULL)) + 8;        // behind the 4 quadrant words
}

// all passes of the staged word at once, kept in LDS behind the culling records: the forward and the backward loop of a
// tile read them from there (one copy of the test in the kernel, one evaluation per tile)
__device__ inline unsigned long long* quadrant_words(float4* srec, int K) {
    return reinterpret_cast<unsigned long long*>(srec + raster_nslot(K) * (R_SLOT + R_CULL));
}
__device__ inline void quadrant_bits_all(const Tile& T, float4* srec, int n, int K, int H, int W) {
    unsigned long long* sq = quadrant_words(srec, K);
    for (int pass = 0; pass * 16 < n; ++pass) {
        const unsigned long long qb = quadrant_bits(T, srec, n, pass, K, H, W);
        if ((threadIdx.x & 63) == 0) sq[pass] = qb;
    }
    __builtin_amdgcn_wave_barrier();
}

// Fused image losses: SilhouetteLoss (L1 / MSE mean against the GT silhouette, modules/loss/silhouette.py:11,22) and
// an L1 depth loss are evaluated where the pixel is produced, so alpha / depth and their gradients never travel
// through HBM.
struct LossArgs {
    const float* gt_sil;      // [B,H,W] or null
    const float* gt_depth;    // [B,H,W] or null
    int sil_mse;              // 0: L1Loss, 1: MSELoss
    float inv_count;          // 1 / (B*H*W): both losses are means
    float* tile_loss;         // fwd out: [B*tiles][2] per-tile sums
    const float* grad_loss;   // bwd in: [2] upstream gradients of the two scalar losses (device); fused kernel: null
    float w_sil, w_dep;       // fused kernel: weights of the two losses in the total
};

__device__ inline float sign0(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }   // torch.sign

// Loss finalisation INSIDE the tile kernel of the training step (it used to be a launch of its own: 7-10 us of latency
// chain): every tile wave publishes its two loss sums and counts itself in on its sample; the wave that arrives last
// on a sample sums that sample's tiles (fixed order) and its Chamfer term (from the per-workgroup sums the scan left),
// publishes the per-sample values and counts the sample in; the wave that completes the last sample writes
//   losses[4] = (silhouette loss, depth loss, w_cd mean_b cd_b + w_sil [0] + w_dep [1], mean_b cd_b).
// Hand-off form (MI355X guide, inter-workgroup visibility): payload stored write-through (sc1), the storing wave
// drains (vmcnt(0)), ONE agent-scope add per workgroup, the last adder reads the payload with sc1 loads.
// -DVPN_STRICT_ORDER: the two arrival adds carry acquire-release semantics at agent scope, i.e. the form the HIP memory
// model asks for (the compiler puts an L2 write-back in front of the add and an L1 invalidate behind it); the default
// relies on the hardware argument above, which the guide lists as a valid form but not as an architectural guarantee.
#ifdef VPN_STRICT_ORDER
#define VPN_ARRIVE_ORDER __ATOMIC_ACQ_REL
#else
#define VPN_ARRIVE_ORDER __ATOMIC_RELAXED
#endif

struct FinArgs {
    int enabled = 0;
    int B = 0;
    int* gcounter = nullptr;            // head of the loss workspace
    float4* persample = nullptr;        // [B]: (sil sum, depth sum, cd_b, tile counter as int)
    const float* sums1 = nullptr;       // [B][g1] per-workgroup sums of dist1 (vpn_chamfer_fwd_ws, fp16 filter) or null
    const float* sums2 = nullptr;       // [B][g2]
    int g1 = 0, g2 = 0, N = 0, M = 0;
    float w1 = 0.f, w2 = 0.f, w_cd = 0.f;
    float* losses = nullptr;            // [4]
    float* loss_b = nullptr;            // [B] or null
    unsigned long long* seed_advance = nullptr;   // device step counter of the caller's sampler seed: += 1 when the step is complete
};

__device__ inline void finalize_sample(const FinArgs& fin, const LossArgs& la, int b, int ntile) {
    const int lane = threadIdx.x & 63;
    const unsigned long long* tl = reinterpret_cast<const unsigned long long*>(la.tile_loss) + (size_t)b * ntile;
    float a0 = 0.f, a1 = 0.f;
    for (int t = lane; t < ntile; t += 64) {
        const unsigned long long v = __hip_atomic_load(tl + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a0 += __uint_as_float((unsigned)v); a1 += __uint_as_float((unsigned)(v >> 32));
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    float cd = 0.f;
    if (fin.sums1) {                     // written by an earlier launch on this stream: plain loads
        float c1 = 0.f, c2 = 0.f;
        for (int i = lane; i < fin.g1; i += 64) c1 += fin.sums1[(size_t)b * fin.g1 + i];
        for (int i = lane; i < fin.g2; i += 64) c2 += fin.sums2[(size_t)b * fin.g2 + i];
        c1 = wave_sum(c1); c2 = wave_sum(c2);
        cd = fin.w1 * (c1 / (float)fin.N) + fin.w2 * (c2 / (float)fin.M);      // chamfer_distance.py:25-28
    }
    int old = 0;
    if (lane == 0) {
        if (fin.loss_b) fin.loss_b[b] = cd;
        float* ps = reinterpret_cast<float*>(fin.persample + b);
        __hip_atomic_store(ps + 0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ps + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ps + 2, cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        old = __hip_atomic_fetch_add(fin.gcounter, 1, VPN_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__builtin_amdgcn_readfirstlane(old) != fin.B - 1) return;
    // every sample is complete
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int i = lane; i < fin.B; i += 64) {
        float* ps = reinterpret_cast<float*>(fin.persample + i);
        t0 += __hip_atomic_load(ps + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t1 += __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(ps + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<int*>(ps + 3), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2);
    if (lane == 0) {
        const float l0 = t0 * la.inv_count, l1 = t1 * la.inv_count, cdm = t2 / (float)fin.B;
        fin.losses[0] = l0; fin.losses[1] = l1;
        fin.losses[2] = fin.w_cd * cdm + la.w_sil * l0 + la.w_dep * l1;
        fin.losses[3] = cdm;
        __hip_atomic_store(fin.gcounter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fin.seed_advance) *fin.seed_advance += 1ull;         // the next step draws fresh surface points
    }
}

// forward composite of the tile's visible primitives: P = prod(1 - a), S0 = sum w, S1 = sum w z
__device__ inline unsigned long long tile_forward(const Tile& T, const float4* __restrict__ rec_b, unsigned long long* __restrict__ mrow,
                                    int words, int K, int H, int W, float4* srec, float inv_sigma, float inv_gamma,
                                    float zref, float P[R_PPL], float S0[R_PPL], float S1[R_PPL], bool masks_ready = false,
                                    bool quad_ready = false, bool has_m = false, unsigned long long m_val = 0ull) {
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) { P[s] = 1.0f; S0[s] = 0.0f; S1[s] = 0.0f; }
    unsigned long long m0 = 0ull;
    for (int w = 0; w < words; ++w) {
        const unsigned long long m = masks_ready ? stage_word(T, rec_b, w, K, H, W, srec, mrow, nullptr, !quad_ready, has_m && w == 0, m_val)
                                                   : stage_word(T, rec_b, w, K, H, W, srec, nullptr, mrow);
        if (w == 0) m0 = m;
        const int n = __builtin_popcountll(m);
        if (!quad_ready) quadrant_bits_all(T, srec, n, K, H, W);          // on: the entry's words are in LDS already
        unsigned long long qb = 0ull;
        for (int j = 0; j < n; ++j) {
            if ((j & 15) == 0) qb = uniform64(quadrant_words(srec, K)[j >> 4]);
            const unsigned qm = (unsigned)(qb >> (4 * (j & 15))) & 15u;
            const float4 q0 = srec[j * R_SLOT], q1 = srec[j * R_SLOT + 1], q2 = srec[j * R_SLOT + 2], q3 = srec[j * R_SLOT + 3];
            auto body = [&](auto kind_c) {
#pragma unroll
                for (int s = 0; s < R_PPL; ++s) {
                    if (!(qm & (1u << s))) continue;            // wave-uniform: the primitive cannot reach this quadrant
                    PixPrim q;
                    eval_prim<decltype(kind_c)::value>(q0, q1, q2, q3, T.px(s), T.py(s), inv_sigma, inv_gamma, zref, q);
                    P[s] *= q.c;
                    S0[s] += q.wgt;
                    S1[s] += q.wgt * q.z;
                }
            };
            if (__builtin_amdgcn_readfirstlane(__float_as_int(q0.w)) == VPN_SPHERE) body(std::integral_constant<int, VPN_SPHERE>{});
            else body(std::integral_constant<int, VPN_CUBOID>{});
        }
        if (words > 1) __builtin_amdgcn_wave_barrier();      // the next word overwrites srec
    }
    return m0;
}

// backward over the tile's visible primitives: per pixel x primitive analytic gradient w.r.t. the 12 ray
// coefficients, summed over the lane's 4 pixels, reduced over the wave and written as 48 contiguous bytes of
// partial[b][k][tile] -- only for the (primitive, tile) pairs of the tile masks; raster_bwd_finish_kernel reads
// exactly those.  No atomics: bitwise reproducible.  `staged`: srec already holds the records of mask word 0 = m0
// (one-pass kernel with K <= 64); mrow: the tile's mask words (stored by an earlier launch, or by this wave's lane 0 in
// its forward half: stage_word takes lane 0's value), or null to repeat the test.
// -DR_DEBUG_LIN (tools/raster_linearity_check.py): for ONE chosen (image, primitive, tile) every lane's twelve sums before
// the wave reduction and, per lane and pixel slot, the backward's intermediates -- to find the operation in which doubling
// the incoming gradient is not exact
#ifdef R_DEBUG_LIN
struct RasterDebugLin { int b, k, tile; float* buf; };
__device__ RasterDebugLin g_dbg_lin = {-1, -1, -1, nullptr};
extern "C" int vpn_debug_raster_lin(int b, int k, int tile, float* buf) {
    RasterDebugLin h{b, k, tile, buf};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_lin), &h, sizeof(h));
}
#endif

__device__ inline void tile_backward(const Tile& T, const float4* __restrict__ rec_b, const unsigned long long* __restrict__ mrow,
                                     int words, int K, int ntile, int H, int W, float4* srec, bool staged,
                                     unsigned long long m0, float inv_sigma,
                                     float inv_gamma, float zref, const float P[R_PPL], const float zbar[R_PPL],
                                     const float invS[R_PPL], const float gAtot[R_PPL], const float gZbar[R_PPL],
                                     float* __restrict__ partial, bool quad_ready = false) {
    const int lane = threadIdx.x & 63;
    for (int w = 0; w < words; ++w) {
        const unsigned long long mw = staged ? m0 : stage_word(T, rec_b, w, K, H, W, srec, mrow, nullptr);
        const int n = __builtin_popcountll(mw);
        if (!staged) quadrant_bits_all(T, srec, n, K, H, W);          // staged: the forward loop left them in LDS
        unsigned long long qb = 0ull;
        for (int j = 0; j < n; ++j) {
            if ((j & 15) == 0) qb = uniform64(quadrant_words(srec, K)[j >> 4]);
            const unsigned qm = (unsigned)(qb >> (4 * (j & 15))) & 15u;
            const float4 q0 = srec[j * R_SLOT], q1 = srec[j * R_SLOT + 1], q2 = srec[j * R_SLOT + 2], q3 = srec[j * R_SLOT + 3];
            const int k = __builtin_amdgcn_readfirstlane(__float_as_int(q1.w));
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = 0.0f;
            auto body = [&](auto kind_c) {
                constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
                for (int s = 0; s < R_PPL; ++s) {
                    if (!(qm & (1u << s))) continue;
                    PixPrim q;
                    eval_prim<KIND>(q0, q1, q2, q3, T.px(s), T.py(s), inv_sigma, inv_gamma, zref, q);
                    // composite backward
                    const float gw = gZbar[s] * (q.z - zbar[s]) * invS[s];
                    float gz = gZbar[s] * q.wgt * invS[s];
                    if (q.ein) gz -= gw * q.wgt * inv_gamma;
                    const float ga = gAtot[s] * (P[s] * R_RCP(q.c)) + gw * q.E;
                    const float gx = q.xin ? ga * q.a * q.c : 0.0f;
                    const float gm2 = -gx * inv_sigma;
                    float go[3], gd[3];
                    prim_backward<KIND>(q0, q, gz, gm2, go, gd);
#ifdef R_DEBUG_LIN
                    if (g_dbg_lin.buf && T.b == g_dbg_lin.b && k == g_dbg_lin.k && T.tile == g_dbg_lin.tile) {
                        float* o = g_dbg_lin.buf + 64 * 16 + (lane * R_PPL + s) * 16;
                        o[0] = gw; o[1] = gz; o[2] = ga; o[3] = gx; o[4] = gm2; o[5] = go[0]; o[6] = go[1]; o[7] = go[2];
                        o[8] = gd[0]; o[9] = gd[1]; o[10] = gd[2]; o[11] = q.wgt; o[12] = q.E; o[13] = q.a; o[14] = gAtot[s]; o[15] = gZbar[s];
                    }
#endif
                    v[0] += go[0]; v[1] += go[1]; v[2] += go[2];
                    v[3] += T.px(s) * gd[0]; v[4] += T.px(s) * gd[1]; v[5] += T.px(s) * gd[2];
                    v[6] += T.py(s) * gd[0]; v[7] += T.py(s) * gd[1]; v[8] += T.py(s) * gd[2];
                    v[9] += gd[0]; v[10] += gd[1]; v[11] += gd[2];
                }
            };
            if (__builtin_amdgcn_readfirstlane(__float_as_int(q0.w)) == VPN_SPHERE) body(std::integral_constant<int, VPN_SPHERE>{});
            else body(std::integral_constant<int, VPN_CUBOID>{});
#ifdef R_DEBUG_LIN
            if (g_dbg_lin.buf && T.b == g_dbg_lin.b && k == g_dbg_lin.k && T.tile == g_dbg_lin.tile)
                for (int i = 0; i < 12; ++i) g_dbg_lin.buf[lane * 16 + i] = v[i];
#endif
            const float tot = reduce12_lds(v, reduce_scratch(srec, K));
            if ((lane & 3) == 0 && (lane >> 2) < 12)
                partial[(((size_t)T.b * K + k) * ntile + T.tile) * 12 + (lane >> 2)] = tot;
        }
        if (words > 1) __builtin_amdgcn_wave_barrier();
    }
}

template <int MODE>   // 0: write alpha/depth images, 1: fused losses (per-tile sums)
__global__ __launch_bounds__(64) void raster_fwd_kernel(const float4* __restrict__ rec,
                                                         unsigned long long* __restrict__ masks,
                                                         const float* __restrict__ cam, int B, int K, int H, int W,
                                                         int tiles_x, int tiles_y, int words, float sigma, float gamma,
                                                         float z_far, float* __restrict__ alpha,
                                                         float* __restrict__ depth, float* __restrict__ aux, LossArgs la) {
    const Tile T = make_tile(H, W, tiles_x, tiles_y, B);
    const int ntile = tiles_x * tiles_y, lane = threadIdx.x & 63;
    const float vz0 = __int_as_float(vzero());
    const float inv_sigma = 1.0f / (sigma + vz0), inv_gamma = 1.0f / (gamma + vz0), zref = cam[T.b * 3] + vz0;
    extern __shared__ __attribute__((aligned(16))) float4 srec[];      // raster_lds_bytes(K)
    float P[R_PPL], S0[R_PPL], S1[R_PPL];
    tile_forward(T, rec + (size_t)T.b * K * R_REC, masks + ((size_t)T.b * ntile + T.tile) * words, words, K, H, W, srec,
                 inv_sigma, inv_gamma, zref, P, S0, S1);
    const size_t hw = (size_t)H * W;
    float lsil = 0.0f, ldep = 0.0f;
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int row = T.row(s), col = T.col(s);
        if (col < W && row < H) {
            const float A = 1.0f - P[s];
            const float S = S0[s] + R_DELTA_S0;
            const float zbar = S1[s] * R_RCP(S);
            const float D = z_far + A * (zbar - z_far);
            const size_t pix = (size_t)row * W + col;
            if (MODE == 0) {
                alpha[T.b * hw + pix] = A;
                depth[T.b * hw + pix] = D;
            } else {
                if (la.gt_sil) { const float e = A - la.gt_sil[T.b * hw + pix]; lsil += la.sil_mse ? e * e : fabsf(e); }
                if (la.gt_depth) ldep += fabsf(D - la.gt_depth[T.b * hw + pix]);
            }
            aux[(T.b * 3 + 0) * hw + pix] = P[s];
            aux[(T.b * 3 + 1) * hw + pix] = zbar;
            aux[(T.b * 3 + 2) * hw + pix] = S;
        }
    }
    if (MODE == 1) {
        lsil = wave_sum(lsil);
        ldep = wave_sum(ldep);
        if (lane == 0) {
            la.tile_loss[((size_t)T.b * ntile + T.tile) * 2 + 0] = lsil;
            la.tile_loss[((size_t)T.b * ntile + T.tile) * 2 + 1] = ldep;
        }
    }
}

template <int MODE>   // 0: incoming gradient images, 1: gradients of the fused losses computed in place
__global__ __launch_bounds__(64, 4) void raster_bwd_kernel(const float4* __restrict__ rec,
                                                         const unsigned long long* __restrict__ masks,
                                                         const float* __restrict__ cam, int B, int K, int H, int W,
                                                         int tiles_x, int tiles_y, int words, float sigma, float gamma,
                                                         float z_far, const float* __restrict__ aux,
                                                         const float* __restrict__ galpha,
                                                         const float* __restrict__ gdepth,
                                                         float* __restrict__ partial, LossArgs la) {
    const Tile T = make_tile(H, W, tiles_x, tiles_y, B);
    const int ntile = tiles_x * tiles_y;
    const unsigned long long* mrow = masks + ((size_t)T.b * ntile + T.tile) * words;
    const float vz0 = __int_as_float(vzero());
    const float inv_sigma = 1.0f / (sigma + vz0), inv_gamma = 1.0f / (gamma + vz0), zref = cam[T.b * 3] + vz0;
    const size_t hw = (size_t)H * W;
    float gAtot[R_PPL], gZbar[R_PPL], P[R_PPL], zbar[R_PPL], invS[R_PPL];
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int row = T.row(s), col = T.col(s);
        gAtot[s] = 0.0f; gZbar[s] = 0.0f; P[s] = 1.0f; zbar[s] = 0.0f; invS[s] = 0.0f;
        if (col < W && row < H) {
            const size_t pix = (size_t)row * W + col;
            P[s] = aux[(T.b * 3 + 0) * hw + pix];
            zbar[s] = aux[(T.b * 3 + 1) * hw + pix];
            invS[s] = R_RCP(aux[(T.b * 3 + 2) * hw + pix]);
            float gA = 0.0f, gD = 0.0f;
            if (MODE == 0) {
                gA = galpha ? galpha[T.b * hw + pix] : 0.0f;
                gD = gdepth ? gdepth[T.b * hw + pix] : 0.0f;
            } else {
                const float A = 1.0f - P[s];
                if (la.gt_sil) {
                    const float e = A - la.gt_sil[T.b * hw + pix];
                    gA = la.grad_loss[0] * la.inv_count * (la.sil_mse ? 2.0f * e : sign0(e));
                }
                if (la.gt_depth) {
                    const float D = z_far + A * (zbar[s] - z_far);
                    gD = la.grad_loss[1] * la.inv_count * sign0(D - la.gt_depth[T.b * hw + pix]);
                }
            }
            gAtot[s] = gA + gD * (zbar[s] - z_far);     // depth = z_far + A (zbar - z_far)
            gZbar[s] = gD * (1.0f - P[s]);
        }
    }
    extern __shared__ __attribute__((aligned(16))) float4 srec[];      // raster_lds_bytes(K)
    tile_backward(T, rec + (size_t)T.b * K * R_REC, mrow, words, K, ntile, H, W, srec, false, 0ull, inv_sigma,
                  inv_gamma, zref, P, zbar, invS, gAtot, gZbar, partial);
}

// Forward AND backward of  total = w_sil * SilhouetteLoss + w_dep * L1(depth)  in one pass over the image (the
// training step, train.py:243-262): the per-pixel state (P, zbar, 1/S) stays in registers between the composite and
// the gradient pass, so the 12 bytes per pixel of `aux` are neither written nor read back, the GT images are read
// once, and there is one launch (and one tail) instead of two.  The gradient partials are those of d total / d(ray
// coefficients) for an upstream gradient of 1: d total is linear in it, raster_bwd_finish_kernel multiplies by the
// actual upstream gradient when backward runs.
// waves per SIMD the register allocation aims at, module-path instantiation.  4 = 107 VGPRs and no scratch; 5 = 96 VGPRs
// with nine spilled dwords (36 B scratch) -- the C2 step times the same with either on one box
// (profiles/r04_c2_waves_ab.txt: 34.9 us both, the kernel 22.3 against 21.9 us); 6 (80 VGPRs) spills in earnest: 63 us.
#ifndef R_TOTAL_WAVES
#define R_TOTAL_WAVES 4
#endif
#ifdef R_EXP_TRACE
// timeline experiment (tools/raster_timeline.py): per tile wave (blockIdx) its start and end on the 100 MHz wall clock,
// the number of visible primitives and the hardware id of the SIMD it ran on
__device__ unsigned long long g_rtrace[65536 * 8];
extern "C" int vpn_debug_raster_trace(void* dst, int nwaves) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rtrace), (size_t)nwaves * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToDevice);
}
#endif
// ENT: the training step's form -- every wave gets its tile entry (K <= 64: one mask word, masks and quadrant words given),
// so the visibility tests, the multi-word loops and the culling records are compiled OUT of that instantiation
#ifndef R_ENT_WAVES
#define R_ENT_WAVES 5
#endif
template <bool ENT>
__global__ __launch_bounds__(64, ENT ? R_ENT_WAVES : R_TOTAL_WAVES) void raster_total_kernel(const float4* __restrict__ rec,
                                                           unsigned long long* __restrict__ masks,
                                                           const float* __restrict__ cam, int B, int K, int H, int W,
                                                           int tiles_x, int tiles_y, int words_arg, float sigma, float gamma,
                                                           float z_far, float* __restrict__ partial, LossArgs la, FinArgs fin,
                                                           const TileEntry* __restrict__ entries) {
    const int words = ENT ? 1 : words_arg;
#ifdef R_EXP_TRACE
    const unsigned long long t_start = wall_clock64();
#endif
    // training step: ONE 48-byte entry (written by the rider of the scan's launch) says which tile this wave takes, which
    // primitives it sees and which quadrants each of them reaches
    const int ntile = tiles_x * tiles_y, lane = threadIdx.x & 63;
    bool quad_ready = false;
    unsigned long long m_entry = 0ull;
    int tile_of_entry = -1;
    uint4 e1q = make_uint4(0u, 0u, 0u, 0u), e2q = e1q;
    if (ENT) {
        const int pt = blockIdx.x / B, eb = blockIdx.x - pt * B;
        const uint4* e = reinterpret_cast<const uint4*>(entries + (size_t)eb * ntile + pt);
        const uint4 e0 = e[0], e1 = e[1], e2 = e[2];
        auto u64 = [](unsigned lo, unsigned hi) {
            return ((unsigned long long)__builtin_amdgcn_readfirstlane(hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(lo);
        };
        tile_of_entry = __builtin_amdgcn_readfirstlane((int)e0.x);
        m_entry = u64(e0.z, e0.w);
        quad_ready = true;
        e1q = e1; e2q = e2;
    }
    const Tile T = make_tile(H, W, tiles_x, tiles_y, B, tile_of_entry);
    const float4* rec_b = rec + (size_t)T.b * K * R_REC;
    unsigned long long* mrow = masks + ((size_t)T.b * ntile + T.tile) * words;
    const float vz0 = __int_as_float(vzero());           // +0.0f in a VGPR: keeps the loop's uniform scalars out of SGPRs
    const float inv_sigma = 1.0f / (sigma + vz0), inv_gamma = 1.0f / (gamma + vz0), zref = cam[T.b * 3] + vz0;
    const size_t hw = (size_t)H * W;
    // GT pixels first: their latency hides behind the composite
    float gs[R_PPL], gd[R_PPL];
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int row = T.row(s), col = T.col(s);
        const bool in = col < W && row < H;
        const size_t pix = in ? (size_t)row * W + col : 0;
        gs[s] = (la.gt_sil && in) ? la.gt_sil[T.b * hw + pix] : 0.0f;
        gd[s] = (la.gt_depth && in) ? la.gt_depth[T.b * hw + pix] : 0.0f;
    }
    extern __shared__ __attribute__((aligned(16))) float4 srec[];      // raster_lds_bytes(K)
    if (ENT && lane == 0) {                                             // every lane holds the same two uint4
        uint4* sq = reinterpret_cast<uint4*>(quadrant_words(srec, K));
        sq[0] = e1q; sq[1] = e2q;
    }
    float P[R_PPL], S0[R_PPL], S1[R_PPL];
    // with entries the tile's mask is known (words == 1 there): stage_word takes it from `m_entry`
    const unsigned long long m0 = tile_forward(T, rec_b, mrow, words, K, H, W, srec, inv_sigma, inv_gamma, zref, P, S0, S1, ENT, quad_ready, ENT, m_entry);
#ifdef R_EXP_TRACE
    const unsigned long long t_fwd = wall_clock64();
#endif
    float gAtot[R_PPL], gZbar[R_PPL], zbar[R_PPL], invS[R_PPL];
    float lsil = 0.0f, ldep = 0.0f;
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int row = T.row(s), col = T.col(s);
        gAtot[s] = 0.0f; gZbar[s] = 0.0f; zbar[s] = 0.0f; invS[s] = 0.0f;
        if (col < W && row < H) {
            const float A = 1.0f - P[s];
            const float S = S0[s] + R_DELTA_S0;
            invS[s] = R_RCP(S);
            zbar[s] = S1[s] * invS[s];
            const float D = z_far + A * (zbar[s] - z_far);
            float gA = 0.0f, gD = 0.0f;
            if (la.gt_sil) {
                const float e = A - gs[s];
                lsil += la.sil_mse ? e * e : fabsf(e);
                gA = la.w_sil * la.inv_count * (la.sil_mse ? 2.0f * e : sign0(e));
            }
            if (la.gt_depth) {
                const float e = D - gd[s];
                ldep += fabsf(e);
                gD = la.w_dep * la.inv_count * sign0(e);
            }
            gAtot[s] = gA + gD * (zbar[s] - z_far);
            gZbar[s] = gD * A;
        } else {
            P[s] = 1.0f;
        }
    }
    lsil = wave_sum(lsil);
    ldep = wave_sum(ldep);
    int arrived = 0;
    if (!fin.enabled) {
        if (lane == 0) {
            la.tile_loss[((size_t)T.b * ntile + T.tile) * 2 + 0] = lsil;
            la.tile_loss[((size_t)T.b * ntile + T.tile) * 2 + 1] = ldep;
        }
    } else if (lane == 0) {
        // publish (write-through), drain, count in: the answer is needed only after the backward half below, which
        // hides the round trip of the add
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(la.tile_loss) + (size_t)T.b * ntile + T.tile,
                           (unsigned long long)__float_as_uint(lsil) | ((unsigned long long)__float_as_uint(ldep) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        arrived = __hip_atomic_fetch_add(reinterpret_cast<int*>(fin.persample + T.b) + 3, 1, VPN_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef R_EXP_TRACE
    const unsigned long long t_loss = wall_clock64();
#endif
    // K > 64: the backward restages word by word from the masks this wave stored (or read) in the forward half -- lane 0
    // reads back its own stores, program order -- instead of repeating the visibility test
    tile_backward(T, rec_b, mrow, words, K, ntile, H, W, srec, ENT || words == 1, m0, inv_sigma, inv_gamma, zref, P, zbar, invS,
                  gAtot, gZbar, partial, quad_ready);
#ifdef R_EXP_TRACE
    const unsigned long long t_bwd = wall_clock64();
#endif
    if (fin.enabled && __builtin_amdgcn_readfirstlane(arrived) == ntile - 1) finalize_sample(fin, la, T.b, ntile);
#ifdef R_EXP_TRACE
    if (lane == 0 && blockIdx.x < 65536) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long* g = g_rtrace + (size_t)blockIdx.x * 8;
        g[0] = t_start; g[1] = wall_clock64(); g[2] = (unsigned long long)__builtin_popcountll(m0); g[3] = hwid;
        g[4] = t_fwd; g[5] = t_loss; g[6] = t_bwd; g[7] = 0;
    }
#endif
}

// one wave per (b,k): raster_finish_wave (vpn_raster_common.h), then write / accumulate the gradient
__global__ __launch_bounds__(256) void raster_bwd_finish_kernel(const float* __restrict__ params,
                                                                const float4* __restrict__ rec, int BK, int K,
                                                                int ntile, int words,
                                                                const unsigned long long* __restrict__ masks,
                                                                const float* __restrict__ partial,
                                                                const float* __restrict__ scale,
                                                                float* __restrict__ gparams, int accumulate) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bk = blockIdx.x * 4 + wave;
    if (bk >= BK) return;
    float r[10];
    raster_finish_wave(params, rec, bk, K, ntile, words, masks, partial, r);
    if (lane != 0) return;
    const float sc = scale ? *scale : 1.0f;
    float* o = gparams + (size_t)bk * VPN_PARAM_STRIDE;
#pragma unroll
    for (int i = 0; i < 10; ++i) o[i] = accumulate ? o[i] + sc * r[i] : sc * r[i];   // accumulate: add to the sampler's gradient
}

// ---------------------------------------------------------------------------------------------------------------
// Loss finalisation: one workgroup per sample sums that sample's tile losses and (optionally) its Chamfer minima
// (chamfer_distance.py:25-28) in a fixed order; the workgroup that arrives last sums the per-sample values in sample
// order and writes
//   losses[0] = silhouette loss, [1] = depth loss, [2] = w_cd * mean_b cd_b + w_sil * [0] + w_dep * [1], [3] = mean_b cd_b.
// Which workgroup is last varies, what it computes does not: bitwise reproducible.  ws = {counter[4] | persample[B][4]}.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ tile_loss, int ntile, int B,
                                                            float inv_count, const float* __restrict__ d1,
                                                            const float* __restrict__ d2, int N, int M, float w1,
                                                            float w2, float w_cd, float w_sil, float w_dep,
                                                            int* __restrict__ counter, float4* __restrict__ persample,
                                                            float* __restrict__ losses, float* __restrict__ loss_b) {
    __shared__ float red[4][4];
    __shared__ int is_last;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (tile_loss) {
        const float2* tl = reinterpret_cast<const float2*>(tile_loss) + (size_t)b * ntile;
        for (int t = threadIdx.x; t < ntile; t += 256) { const float2 v = tl[t]; acc[0] += v.x; acc[1] += v.y; }
    }
    if (d1) {
        // fixed order: thread t owns elements t, t + 256, ... (float4 groups where the row is 16-byte aligned); the
        // loads of a batch of 8 are issued together, then summed
        auto row_sum = [&](const float* __restrict__ a, int n) -> float {
            float s = 0.f;
            if ((n & 3) == 0) {
                const float4* a4 = reinterpret_cast<const float4*>(a);
                const int n4 = n >> 2;
                for (int i = threadIdx.x; i < n4; i += 8 * 256) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (i + u * 256 < n4) ? a4[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                }
            } else {
                for (int i = threadIdx.x; i < n; i += 256) s += a[i];
            }
            return s;
        };
        acc[2] = row_sum(d1 + (size_t)b * N, N);
        acc[3] = row_sum(d2 + (size_t)b * M, M);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const float cd = d1 ? w1 * (s[2] / (float)N) + w2 * (s[3] / (float)M) : 0.0f;
        if (loss_b) loss_b[b] = cd;
        float* ps = reinterpret_cast<float*>(persample + b);
        __hip_atomic_store(ps + 0, s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1 stores: past L2's dirty lines
        __hip_atomic_store(ps + 1, s[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ps + 2, cd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == B - 1;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!is_last) return;
    float t[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < B; i += 256) {
        const float* ps = reinterpret_cast<const float*>(persample + i);
        t[0] += __hip_atomic_load(ps + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[1] += __hip_atomic_load(ps + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[2] += __hip_atomic_load(ps + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                         // red is reused
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float s = wave_sum(t[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const float l0 = s[0] * inv_count, l1 = s[1] * inv_count, cd = s[2] / (float)B;
        losses[0] = l0; losses[1] = l1;
        losses[2] = w_cd * cd + w_sil * l0 + w_dep * l1;
        losses[3] = cd;
        __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
    }
}

struct Grid { dim3 g; int tiles_x, tiles_y, ntile, words; };
static inline Grid raster_grid(int B, int K, int H, int W) {
    Grid G;
    G.tiles_x = (W + R_TW - 1) / R_TW; G.tiles_y = (H + R_TH - 1) / R_TH;
    G.ntile = G.tiles_x * G.tiles_y; G.words = (K + 63) / 64;
    G.g = dim3((unsigned)G.ntile * (unsigned)B);
    return G;
}

}  // namespace vpn

using namespace vpn;

static int raster_check(const void* params, const void* kinds, const void* cam, int B, int K, int H, int W,
                        float sigma, float gamma) {
    if (!params || !kinds || !cam) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f) || !(gamma > 0.f)) return VPN_E_BADARG;
    if (K > VPN_MAX_PRIMS || B > 65535 || (H + R_TH - 1) / R_TH > 65535) return VPN_E_TOOBIG;
    return 0;
}

// records = [B*K][R_REC] float4, then the tile masks [B*ntile][words] uint64
static inline size_t rec_bytes(int B, int K) { return (size_t)B * K * R_REC * sizeof(float4); }
extern "C" size_t vpn_raster_records_size(int B, int K, int H, int W) {
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    const Grid G = raster_grid(B, K, H, W);
    return rec_bytes(B, K) + (size_t)B * G.ntile * G.words * sizeof(unsigned long long);
}
static inline unsigned long long* masks_of(void* records, int B, int K) {
    return reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(records) + rec_bytes(B, K));
}
static inline const unsigned long long* masks_of(const void* records, int B, int K) {
    return reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(records) + rec_bytes(B, K));
}

// the record kernel lives in sampler.hip: records written by the sampler's forward launch (hot path) and by this
// stand-alone launch come from ONE compiled definition, so the pose the fused backward reads back is the sampler's own
namespace vpn { int launch_raster_prep(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H, int W,
                                       float sigma, void* records, int* zero_me, hipStream_t s); }
static int launch_prep(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H, int W,
                       float sigma, void* records, int* zero_me, hipStream_t s) {
    return vpn::launch_raster_prep(params, kinds, cam, B, K, H, W, sigma, records, zero_me, s);
}

extern "C" int vpn_raster_fwd(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                              int W, float sigma, float gamma, float z_far, float* alpha, float* depth, float* aux,
                              void* records, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!alpha || !depth || !aux || !records) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0) return VPN_E_BADARG;
    if ((rc = launch_prep(params, kinds, cam, B, K, H, W, sigma, records, nullptr, (hipStream_t)stream))) return rc;
    const Grid G = raster_grid(B, K, H, W);
    VPN_LAUNCH(raster_fwd_kernel<0>, G.g, dim3(64), raster_lds_bytes(K), (hipStream_t)stream, (const float4*)records,
               masks_of(records, B, K), cam, B, K, H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, alpha,
               depth, aux, LossArgs{});
    VPN_LAUNCH_CHECK();
    return 0;
}

// loss workspace = {arrival counter (16 B) | per-sample sums [B] float4 | per-tile sums [B*ntile] float2}
static inline size_t loss_head_bytes(int B) { return 16 + (size_t)B * sizeof(float4); }
extern "C" size_t vpn_raster_loss_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const Grid G = raster_grid(B, 1, H, W);
    return loss_head_bytes(B) + (size_t)B * G.ntile * 2 * sizeof(float);
}
static inline float* tile_loss_of(void* ws, int B) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + loss_head_bytes(B)); }

static int launch_finalize(void* loss_ws, bool with_tiles, int B, int H, int W, const float* d1, const float* d2, int N,
                           int M, float w1, float w2, float w_cd, float w_sil, float w_dep, float* losses, float* loss_b,
                           hipStream_t s) {
    const Grid G = raster_grid(B, 1, H > 0 ? H : 1, W > 0 ? W : 1);
    const float inv_count = with_tiles ? 1.0f / ((float)B * (float)H * (float)W) : 0.0f;
    VPN_LAUNCH(loss_finalize_kernel, dim3(B), dim3(256), 0, s, with_tiles ? (const float*)tile_loss_of(loss_ws, B) : nullptr,
               G.ntile, B, inv_count, d1, d2, N, M, w1, w2, w_cd, w_sil, w_dep, (int*)loss_ws,
               reinterpret_cast<float4*>(reinterpret_cast<char*>(loss_ws) + 16), losses, loss_b);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_raster_loss_fwd(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                                   int W, float sigma, float gamma, float z_far, const float* gt_sil,
                                   const float* gt_depth, int sil_mse, float* aux, void* records, void* loss_ws,
                                   float* losses, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!aux || !records || !loss_ws || !losses) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0 || ((uintptr_t)loss_ws & 15) != 0) return VPN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = launch_prep(params, kinds, cam, B, K, H, W, sigma, records, (int*)loss_ws, s))) return rc;
    const Grid G = raster_grid(B, K, H, W);
    LossArgs la{gt_sil, gt_depth, sil_mse, 1.0f / ((float)B * (float)H * (float)W), tile_loss_of(loss_ws, B), nullptr, 0.f, 0.f};
    VPN_LAUNCH(raster_fwd_kernel<1>, G.g, dim3(64), raster_lds_bytes(K), s, (const float4*)records, masks_of(records, B, K), cam, B, K,
               H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, (float*)nullptr, (float*)nullptr, aux, la);
    VPN_LAUNCH_CHECK();
    return launch_finalize(loss_ws, true, B, H, W, nullptr, nullptr, 0, 0, 0.f, 0.f, 0.f, 1.0f, 1.0f, losses, nullptr, s);
}

extern "C" size_t vpn_raster_bwd_workspace(int B, int K, int H, int W) {
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    const Grid G = raster_grid(B, K, H, W);
    return (size_t)B * G.ntile * K * 12 * sizeof(float);
}

static int launch_finish(const float* params, const float* cam, int B, int K, int H, int W, const void* records,
                         const void* workspace, const float* scale, float* grad_params, int accumulate, hipStream_t s) {
    const Grid G = raster_grid(B, K, H, W);
    const int BK = B * K;
    VPN_LAUNCH(raster_bwd_finish_kernel, dim3((BK + 3) / 4), dim3(256), 0, s, params, (const float4*)records, BK, K, G.ntile, G.words,
               masks_of(records, B, K), (const float*)workspace, scale, grad_params, accumulate);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_raster_bwd(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                              int W, float sigma, float gamma, float z_far, const float* aux, const void* records,
                              const float* grad_alpha, const float* grad_depth, void* workspace,
                              float* grad_params, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!aux || !records || !workspace || !grad_params) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0 || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    const Grid G = raster_grid(B, K, H, W);
    VPN_LAUNCH(raster_bwd_kernel<0>, G.g, dim3(64), raster_lds_bytes(K), (hipStream_t)stream, (const float4*)records, masks_of(records, B, K),
               cam, B, K, H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, aux, grad_alpha, grad_depth,
               (float*)workspace, LossArgs{});
    VPN_LAUNCH_CHECK();
    return launch_finish(params, cam, B, K, H, W, records, workspace, nullptr, grad_params, 0, (hipStream_t)stream);
}

extern "C" int vpn_raster_loss_bwd(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                                   int W, float sigma, float gamma, float z_far, const float* aux, const void* records,
                                   const float* gt_sil, const float* gt_depth, int sil_mse, const float* grad_losses,
                                   void* workspace, float* grad_params, int accumulate, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!aux || !records || !grad_losses || !workspace || !grad_params) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0 || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    const Grid G = raster_grid(B, K, H, W);
    LossArgs la{gt_sil, gt_depth, sil_mse, 1.0f / ((float)B * (float)H * (float)W), nullptr, grad_losses, 0.f, 0.f};
    VPN_LAUNCH(raster_bwd_kernel<1>, G.g, dim3(64), raster_lds_bytes(K), (hipStream_t)stream, (const float4*)records, masks_of(records, B, K),
               cam, B, K, H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, aux, (const float*)nullptr,
               (const float*)nullptr, (float*)workspace, la);
    VPN_LAUNCH_CHECK();
    return launch_finish(params, cam, B, K, H, W, records, workspace, nullptr, grad_params, accumulate, (hipStream_t)stream);
}

extern "C" int vpn_raster_total_fwd(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                                    int W, float sigma, float gamma, float z_far, const float* gt_sil,
                                    const float* gt_depth, int sil_mse, float w_sil, float w_dep, void* records,
                                    void* loss_ws, void* workspace, int records_ready, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!records || !loss_ws || !workspace) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0 || ((uintptr_t)workspace & 15) != 0 || ((uintptr_t)loss_ws & 15) != 0) return VPN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (!records_ready && (rc = launch_prep(params, kinds, cam, B, K, H, W, sigma, records, (int*)loss_ws, s))) return rc;
    const Grid G = raster_grid(B, K, H, W);
    LossArgs la{gt_sil, gt_depth, sil_mse, 1.0f / ((float)B * (float)H * (float)W), tile_loss_of(loss_ws, B), nullptr, w_sil, w_dep};
    VPN_LAUNCH_AS("raster_total_kernel", raster_total_kernel<false>, G.g, dim3(64), raster_lds_bytes(K), s, (const float4*)records, masks_of(records, B, K), cam, B, K,
               H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, (float*)workspace, la, FinArgs{}, (const TileEntry*)nullptr);
    VPN_LAUNCH_CHECK();
    return 0;
}

namespace vpn { int chamfer_wgsums(const void* workspace, size_t workspace_bytes, int B, int N, int M, const float** sums1, int* g1,
                                   const float** sums2, int* g2); }

extern "C" int vpn_raster_total_fwd_fin(const float* params, const int32_t* kinds, const float* cam, int B, int K, int H,
                                        int W, float sigma, float gamma, float z_far, const float* gt_sil,
                                        const float* gt_depth, int sil_mse, float w_sil, float w_dep, void* records,
                                        void* loss_ws, void* workspace, int records_ready,
                                        const void* chamfer_ws, size_t chamfer_ws_bytes, int N, int M, float cd_w1,
                                        float cd_w2, float w_cd, float* losses, float* loss_b, uint64_t* seed_advance,
                                        const void* tile_order, void* stream) {
    int rc = raster_check(params, kinds, cam, B, K, H, W, sigma, gamma);
    if (rc) return rc;
    if (!records || !loss_ws || !workspace || !losses) return VPN_E_BADARG;
    if (((uintptr_t)records & 15) != 0 || ((uintptr_t)workspace & 15) != 0 || ((uintptr_t)loss_ws & 15) != 0) return VPN_E_BADARG;
    FinArgs fin;
    fin.enabled = 1; fin.B = B;
    fin.gcounter = (int*)loss_ws;
    fin.persample = reinterpret_cast<float4*>(reinterpret_cast<char*>(loss_ws) + 16);
    if (chamfer_ws) {
        if (N <= 0 || M <= 0) return VPN_E_BADARG;
        if ((rc = vpn::chamfer_wgsums(chamfer_ws, chamfer_ws_bytes, B, N, M, &fin.sums1, &fin.g1, &fin.sums2, &fin.g2))) return rc;
        fin.N = N; fin.M = M; fin.w1 = cd_w1; fin.w2 = cd_w2; fin.w_cd = w_cd;
    }
    fin.losses = losses; fin.loss_b = loss_b; fin.seed_advance = (unsigned long long*)seed_advance;
    hipStream_t s = (hipStream_t)stream;
    if (!records_ready && (rc = launch_prep(params, kinds, cam, B, K, H, W, sigma, records, (int*)loss_ws, s))) return rc;
    const Grid G = raster_grid(B, K, H, W);
    if (tile_order && (!records_ready || G.words != 1 || ((uintptr_t)tile_order & 15) != 0)) return VPN_E_BADARG;   // entries come with the masks, K <= 64
    LossArgs la{gt_sil, gt_depth, sil_mse, 1.0f / ((float)B * (float)H * (float)W), tile_loss_of(loss_ws, B), nullptr, w_sil, w_dep};
    if (tile_order)
        VPN_LAUNCH_AS("raster_total_kernel", raster_total_kernel<true>, G.g, dim3(64), raster_lds_bytes(K), s, (const float4*)records, masks_of(records, B, K), cam, B, K,
                      H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, (float*)workspace, la, fin, (const TileEntry*)tile_order);
    else
        VPN_LAUNCH_AS("raster_total_kernel", raster_total_kernel<false>, G.g, dim3(64), raster_lds_bytes(K), s, (const float4*)records, masks_of(records, B, K), cam, B, K,
                      H, W, G.tiles_x, G.tiles_y, G.words, sigma, gamma, z_far, (float*)workspace, la, fin, (const TileEntry*)nullptr);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_loss_finalize(void* loss_ws, int B, int H, int W, const float* dist1, const float* dist2, int N, int M,
                                 float cd_w1, float cd_w2, float w_cd, float w_sil, float w_dep, float* losses,
                                 float* loss_b, void* stream) {
    if (!loss_ws || !losses || B <= 0) return VPN_E_BADARG;
    if (((uintptr_t)loss_ws & 15) != 0) return VPN_E_BADARG;
    if ((dist1 == nullptr) != (dist2 == nullptr)) return VPN_E_BADARG;
    if (dist1 && (N <= 0 || M <= 0)) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    return launch_finalize(loss_ws, H > 0 && W > 0, B, H, W, dist1, dist2, N, M, cd_w1, cd_w2, w_cd, w_sil, w_dep, losses, loss_b,
                           (hipStream_t)stream);
}

extern "C" int vpn_raster_total_bwd(const float* params, const float* cam, int B, int K, int H, int W,
                                    const void* records, const void* workspace, const float* grad_total,
                                    float* grad_params, int accumulate, void* stream) {
    if (!params || !cam || !records || !workspace || !grad_params) return VPN_E_BADARG;
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return VPN_E_BADARG;
    if (K > VPN_MAX_PRIMS || B > 65535) return VPN_E_TOOBIG;
    return launch_finish(params, cam, B, K, H, W, records, workspace, grad_total, grad_params, accumulate, (hipStream_t)stream);
}
