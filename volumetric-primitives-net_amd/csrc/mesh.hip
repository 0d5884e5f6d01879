// mesh.hip — the triangle-mesh path of the render surface (gfx950): meshes that carry no primitives.
//
// Replaces, for such meshes (reference file:line): the DIBRenderer call of VertexRenderer.render
// (modules/render/vertex_renderer.py:14-26: vertices (P,3) + faces (F,3) -> alpha), as SilhouetteLoss.forward
// (modules/loss/silhouette.py:13-23) and train_sphere.py:128 use it, and kaolin's TriangleMesh.sample
// (train_sphere.py:76: area-weighted surface samples of the deformed 386-vertex sphere).  Both live in kaolin, which is
// absent: the arithmetic here follows the repo's own specification oracle.vpn_oracle.mesh_raster / mesh_sample
// (parity unpinned), the camera is the one of the primitive raster (vpn_raster_common.h make_camera).
//
// Raster: one wavefront per 16x16 pixel tile (the layout of raster.hip: lane = one pixel in each 8x8 quadrant).  The
// wave walks the face list 64 faces at a time -- lane = face: gather its three projected corners, test the face's
// bounding box (inflated by the reach of the soft edge) against the tile, ballot -- stages the visible faces in LDS and
// evaluates them for its 256 pixels: squared distance to the nearest edge SEGMENT, inside test by the three edge
// functions, a = sigmoid(+-d2 / sigma), alpha = 1 - prod (1 - a).  No face binning pass and no per-tile lists: at the
// reference's sizes (F = 768 for train_sphere.py, <= 16128 for C5 meshes) the scan of the face list is 12 .. 252
// iterations of a wave per tile.  Backward: the same walk; per (tile, face) the gradient of the six projected
// coordinates is reduced over the wave and added (atomics: a face touches a handful of tiles) to the projected
// vertices' gradient, which a per-vertex kernel carries through the projection.
#include "vpn_raster_common.h"

namespace vpn {

constexpr float MR_NEAR = 1e-3f;          // faces with a corner closer to the camera plane are not drawn (oracle MESH_NEAR)
constexpr float MR_X_CLAMP = 80.0f;
constexpr float MR_X_CUT = 16.0f;         // outside a face's box inflated by sqrt(X_CUT sigma) its coverage is < 1.2e-7: skipped
constexpr int MR_LDS_F4 = 64 * 2;         // staged faces per pass: (ax, ay, bx, by), (cx, cy, face, -)
constexpr float MR_MIN_AREA2 = 1e-12f;    // twice the NDC area below which a face has no inside (oracle MESH_MIN_AREA2): 1e-8 of a 256^2 pixel

// ---- projection: (x, y) in NDC (y in [-1,1] over the image height, +x right, +y up), z = depth along the optical axis
__global__ __launch_bounds__(256) void mesh_project_kernel(const float* __restrict__ verts, const float* __restrict__ cam,
                                                           int P, float4* __restrict__ proj) {
    __shared__ Camera C;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) C = make_camera(cam + b * 3);
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const F3 v = ld3(verts + ((size_t)b * P + p) * 3);
    const float rx = v.x - C.eye[0], ry = v.y - C.eye[1], rz = v.z - C.eye[2];
    const float xc = rx * C.right[0] + ry * C.right[1] + rz * C.right[2];
    const float yc = rx * C.up[0] + ry * C.up[1] + rz * C.up[2];
    const float zc = rx * C.fwd[0] + ry * C.fwd[1] + rz * C.fwd[2];
    const float zs = zc > MR_NEAR ? zc : 1.0f;
    const float inv = 1.0f / (zs * R_TAN_HALF_FOV);
    proj[(size_t)b * P + p] = make_float4(xc * inv, yc * inv, zc, 0.0f);
}

// d total / d verts from d total / d (x_ndc, y_ndc) of the projected vertices
__global__ __launch_bounds__(256) void mesh_project_bwd_kernel(const float* __restrict__ verts, const float* __restrict__ cam,
                                                               int P, const float* __restrict__ gproj,
                                                               float* __restrict__ gverts) {
    __shared__ Camera C;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) C = make_camera(cam + b * 3);
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const F3 v = ld3(verts + ((size_t)b * P + p) * 3);
    const float rx = v.x - C.eye[0], ry = v.y - C.eye[1], rz = v.z - C.eye[2];
    const float xc = rx * C.right[0] + ry * C.right[1] + rz * C.right[2];
    const float yc = rx * C.up[0] + ry * C.up[1] + rz * C.up[2];
    const float zc = rx * C.fwd[0] + ry * C.fwd[1] + rz * C.fwd[2];
    float gx = gproj[((size_t)b * P + p) * 2], gy = gproj[((size_t)b * P + p) * 2 + 1];
    float gxc = 0.f, gyc = 0.f, gzc = 0.f;
    if (zc > MR_NEAR) {                                  // x = xc / (zc th), y = yc / (zc th)
        const float inv = 1.0f / (zc * R_TAN_HALF_FOV);
        gxc = gx * inv; gyc = gy * inv;
        gzc = -(gx * xc + gy * yc) * inv / zc;
    }
    st3(gverts + ((size_t)b * P + p) * 3, gxc * C.right[0] + gyc * C.up[0] + gzc * C.fwd[0],
        gxc * C.right[1] + gyc * C.up[1] + gzc * C.fwd[1], gxc * C.right[2] + gyc * C.up[2] + gzc * C.fwd[2]);
}

struct MTile {
    int b, c0, r0;
    float x[2], y[2];          // NDC of this lane's pixel centre in quadrant column / row 0, 1
    int col[2], row[2];
    float tx0, tx1, ty0, ty1;  // the tile's rectangle in NDC
};

__device__ inline MTile mesh_tile(int H, int W, int tiles_x, int B) {
    MTile T;
    const int lane = threadIdx.x & 63;
    const int id = blockIdx.x, pt = id / B;
    T.b = id - pt * B;
    const int ty = pt / tiles_x, tx = pt - ty * tiles_x;
    T.c0 = tx * R_TW; T.r0 = ty * R_TH;
    const float ar = (float)W / (float)H;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        T.col[h] = T.c0 + 8 * h + (lane & 7);
        T.row[h] = T.r0 + 8 * h + (lane >> 3);
        T.x[h] = ((2.0f * ((float)T.col[h] + 0.5f) / (float)W) - 1.0f) * ar;
        T.y[h] = 1.0f - (2.0f * ((float)T.row[h] + 0.5f) / (float)H);
    }
    T.tx0 = ((2.0f * (float)T.c0 / (float)W) - 1.0f) * ar;
    T.tx1 = ((2.0f * (float)(T.c0 + R_TW) / (float)W) - 1.0f) * ar;
    T.ty1 = 1.0f - (2.0f * (float)T.r0 / (float)H);
    T.ty0 = 1.0f - (2.0f * (float)(T.r0 + R_TH) / (float)H);
    return T;
}

// faces [f0, f0 + 64) of the list: lane = face; visible ones are staged in `sface` (slot = number of visible faces
// below the lane).  Returns the wave-uniform mask.
__device__ inline unsigned long long stage_faces(const MTile& T, const float4* __restrict__ proj_b, const int32_t* __restrict__ faces,
                                                 int f0, int F, int P, float reach, float4* sface) {
    const int lane = threadIdx.x & 63;
    const int f = f0 + lane;
    bool vis = false;
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), Bv = A, C = A;
    if (f < F) {
        // vertex indices are data: clamped, so that a bad face can neither fault nor reach another sample's vertices
        const int ia = min(max(faces[f * 3], 0), P - 1), ib = min(max(faces[f * 3 + 1], 0), P - 1), ic = min(max(faces[f * 3 + 2], 0), P - 1);
        A = proj_b[ia]; Bv = proj_b[ib]; C = proj_b[ic];
        const bool ok = A.z > MR_NEAR && Bv.z > MR_NEAR && C.z > MR_NEAR;
        const float x0 = fminf(fminf(A.x, Bv.x), C.x) - reach, x1 = fmaxf(fmaxf(A.x, Bv.x), C.x) + reach;
        const float y0 = fminf(fminf(A.y, Bv.y), C.y) - reach, y1 = fmaxf(fmaxf(A.y, Bv.y), C.y) + reach;
        vis = ok && x0 <= T.tx1 && x1 >= T.tx0 && y0 <= T.ty1 && y1 >= T.ty0;
    }
    const unsigned long long m = __ballot(vis);
    if (vis) {
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        sface[slot * 2] = make_float4(A.x, A.y, Bv.x, Bv.y);
        sface[slot * 2 + 1] = make_float4(C.x, C.y, __int_as_float(f), 0.0f);
    }
    __builtin_amdgcn_wave_barrier();
    return m;
}

// squared distance from (px,py) to segment (x0,y0)-(x1,y1); t = parameter of the closest point, (qx,qy) = p - closest
__device__ inline float seg_d2(float px, float py, float x0, float y0, float x1, float y1, float& t, float& qx, float& qy) {
    const float ex = x1 - x0, ey = y1 - y0, wx = px - x0, wy = py - y0;
    const float l2 = fmaxf(ex * ex + ey * ey, 1e-20f);
    t = __builtin_amdgcn_fmed3f((wx * ex + wy * ey) / l2, 0.0f, 1.0f);
    qx = wx - t * ex; qy = wy - t * ey;
    return qx * qx + qy * qy;
}

struct FaceEval { float a, c, d2, t, qx, qy; int seg; bool inside, live; };

__device__ inline FaceEval eval_face(float px, float py, const float4 f0, const float4 f1, float inv_sigma) {
    FaceEval r;
    const float ax = f0.x, ay = f0.y, bx = f0.z, by = f0.w, cx = f1.x, cy = f1.y;
    float t0, q0x, q0y, t1, q1x, q1y, t2, q2x, q2y;
    const float d0 = seg_d2(px, py, ax, ay, bx, by, t0, q0x, q0y);
    const float d1 = seg_d2(px, py, bx, by, cx, cy, t1, q1x, q1y);
    const float d2 = seg_d2(px, py, cx, cy, ax, ay, t2, q2x, q2y);
    // nearest of the three segments by two rounds of selects (written as `if`s that also set the segment number the
    // compiler turned the choice into a nine-dword table in scratch memory indexed by it)
    const bool b1 = d1 < d0;
    const float dm = b1 ? d1 : d0, tm = b1 ? t1 : t0, qmx = b1 ? q1x : q0x, qmy = b1 ? q1y : q0y;
    const bool b2 = d2 < dm;
    r.d2 = b2 ? d2 : dm; r.t = b2 ? t2 : tm; r.qx = b2 ? q2x : qmx; r.qy = b2 ? q2y : qmy;
    r.seg = b2 ? 2 : (b1 ? 1 : 0);
    const float e0 = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    const float e1 = (cx - bx) * (py - by) - (cy - by) * (px - bx);
    const float e2 = (ax - cx) * (py - cy) - (ay - cy) * (px - cx);
    // e0 + e1 + e2 = twice the signed area, whatever p: a face of (next to) no area has no inside -- with all three
    // edge functions 0 it used to be "inside" for every pixel and painted alpha = 1 over every tile its box touched
    r.inside = ((e0 >= 0.f && e1 >= 0.f && e2 >= 0.f) || (e0 <= 0.f && e1 <= 0.f && e2 <= 0.f)) && fabsf((e0 + e1) + e2) > MR_MIN_AREA2;
    const float xr = (r.inside ? r.d2 : -r.d2) * inv_sigma;
    r.live = fabsf(xr) <= MR_X_CLAMP;
    const float x = __builtin_amdgcn_fmed3f(xr, -MR_X_CLAMP, MR_X_CLAMP);
    const float en = __expf(-x);
    r.a = 1.0f / (1.0f + en);
    r.c = en * r.a;                                    // 1 - a without cancellation
    return r;
}

__global__ __launch_bounds__(64) void mesh_raster_fwd_kernel(const float4* __restrict__ proj, const int32_t* __restrict__ faces,
                                                             int B, int P, int F, int H, int W, int tiles_x, float sigma,
                                                             float* __restrict__ alpha) {
    __shared__ __attribute__((aligned(16))) float4 sface[MR_LDS_F4];
    const MTile T = mesh_tile(H, W, tiles_x, B);
    const float4* proj_b = proj + (size_t)T.b * P;
    const float inv_sigma = 1.0f / sigma, reach = sqrtf(MR_X_CUT * sigma);
    float Pc[R_PPL] = {1.0f, 1.0f, 1.0f, 1.0f};
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int n = __builtin_popcountll(stage_faces(T, proj_b, faces, f0, F, P, reach, sface));
        for (int j = 0; j < n; ++j) {
            const float4 a0 = sface[j * 2], a1 = sface[j * 2 + 1];
#pragma unroll
            for (int s = 0; s < R_PPL; ++s) Pc[s] *= eval_face(T.x[s & 1], T.y[s >> 1], a0, a1, inv_sigma).c;
        }
        __builtin_amdgcn_wave_barrier();                // the next pass overwrites sface
    }
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int col = T.col[s & 1], row = T.row[s >> 1];
        if (col < W && row < H) alpha[((size_t)T.b * H + row) * W + col] = 1.0f - Pc[s];
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) void mesh_raster_bwd_kernel(const float4* __restrict__ proj, const int32_t* __restrict__ faces,
                                                             int B, int P, int F, int H, int W, int tiles_x, float sigma,
                                                             const float* __restrict__ alpha, const float* __restrict__ galpha,
                                                             float* __restrict__ gproj) {
    __shared__ __attribute__((aligned(16))) float4 sface[MR_LDS_F4];
    const MTile T = mesh_tile(H, W, tiles_x, B);
    const int lane = threadIdx.x & 63;
    const float4* proj_b = proj + (size_t)T.b * P;
    const float inv_sigma = 1.0f / sigma, reach = sqrtf(MR_X_CUT * sigma);
    float Pc[R_PPL], gA[R_PPL];
    bool any = false;
#pragma unroll
    for (int s = 0; s < R_PPL; ++s) {
        const int col = T.col[s & 1], row = T.row[s >> 1];
        const bool in = col < W && row < H;
        const size_t pix = in ? ((size_t)T.b * H + row) * W + col : 0;
        Pc[s] = in ? 1.0f - alpha[pix] : 1.0f;
        gA[s] = in ? galpha[pix] : 0.0f;
        any |= gA[s] != 0.0f;
    }
    if (!__ballot(any)) return;                          // no gradient reaches this tile
    float* gp = gproj + (size_t)T.b * P * 2;
    for (int f0 = 0; f0 < F; f0 += 64) {
        const int n = __builtin_popcountll(stage_faces(T, proj_b, faces, f0, F, P, reach, sface));
        for (int j = 0; j < n; ++j) {
            const float4 a0 = sface[j * 2], a1 = sface[j * 2 + 1];
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < R_PPL; ++s) {
                const FaceEval e = eval_face(T.x[s & 1], T.y[s >> 1], a0, a1, inv_sigma);
                // alpha = 1 - prod c_f: d alpha / d a_f = P / c_f; a = sigmoid(x): da/dx = a c; x = +-d2 / sigma
                const float gx = e.live ? gA[s] * (Pc[s] * __builtin_amdgcn_rcpf(e.c)) * e.a * e.c : 0.0f;
                const float gd2 = (e.inside ? gx : -gx) * inv_sigma;
                // d2 = |p - closest|^2 on segment (u -> w): d d2/du = -2 q (1 - t), d d2/dw = -2 q t
                const float gu = -2.0f * gd2 * (1.0f - e.t), gw = -2.0f * gd2 * e.t;
                const float gux = gu * e.qx, guy = gu * e.qy, gwx = gw * e.qx, gwy = gw * e.qy;
                // segment 0 = (a,b), 1 = (b,c), 2 = (c,a)
                v[0] += e.seg == 0 ? gux : (e.seg == 2 ? gwx : 0.f); v[1] += e.seg == 0 ? guy : (e.seg == 2 ? gwy : 0.f);
                v[2] += e.seg == 1 ? gux : (e.seg == 0 ? gwx : 0.f); v[3] += e.seg == 1 ? guy : (e.seg == 0 ? gwy : 0.f);
                v[4] += e.seg == 2 ? gux : (e.seg == 1 ? gwx : 0.f); v[5] += e.seg == 2 ? guy : (e.seg == 1 ? gwy : 0.f);
            }
            const float tot = wave_reduce16(v);          // lane L: wave total of value L >> 2
            const int f = __builtin_amdgcn_readfirstlane(__float_as_int(a1.z));
            if ((lane & 3) == 0 && lane < 24 && tot != 0.0f) {
                const int which = lane >> 2;             // 0..5 = (a.x, a.y, b.x, b.y, c.x, c.y)
                const int vi = min(max(faces[f * 3 + (which >> 1)], 0), P - 1);
                atomicAdd(gp + vi * 2 + (which & 1), tot);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- area-weighted surface samples (train_sphere.py:76)
// one workgroup per mesh: face areas, inclusive prefix sums in fp64 (what torch.cumsum does for fp32 on the CPU),
// rounded to fp32
__global__ __launch_bounds__(1024) void mesh_cdf_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
                                                        int P, int F, float* __restrict__ cdf) {
    __shared__ double wsum[16];
    __shared__ double carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vb = verts + (size_t)b * P * 3;
    if (tid == 0) carry = 0.0;
    __syncthreads();
    for (int f0 = 0; f0 < F; f0 += 1024) {
        const int f = f0 + tid;
        double area = 0.0;
        if (f < F) {
            const int ia = min(max(faces[f * 3], 0), P - 1), ib = min(max(faces[f * 3 + 1], 0), P - 1), ic = min(max(faces[f * 3 + 2], 0), P - 1);
            const F3 a = ld3(vb + ia * 3), bb = ld3(vb + ib * 3), c = ld3(vb + ic * 3);
            const float ux = bb.x - a.x, uy = bb.y - a.y, uz = bb.z - a.z, wx = c.x - a.x, wy = c.y - a.y, wz = c.z - a.z;
            const float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
            area = (double)(0.5f * sqrtf(nx * nx + ny * ny + nz * nz));
        }
        double s = area;                                 // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        double base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (f < F) cdf[(size_t)b * F + f] = (float)(base + s);
        __syncthreads();
        if (tid == 1023) carry = base + s;
        __syncthreads();
    }
}

// one thread per sample point: Philox draw -> face by binary search in the cdf -> barycentric point
__global__ __launch_bounds__(256) void mesh_sample_fwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
                                                              const float* __restrict__ cdf, const float* __restrict__ u,
                                                              uint64_t seed, uint64_t mesh_base, int P, int F, int n,
                                                              float* __restrict__ points, int32_t* __restrict__ face_idx,
                                                              float* __restrict__ bary) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float uu[3];
    if (u) { const float* ub = u + ((size_t)b * n + i) * 3; uu[0] = ub[0]; uu[1] = ub[1]; uu[2] = ub[2]; }
    else philox_uniform3(seed, mesh_base + (uint64_t)b, 0xFFFFFFFFu, (uint32_t)i, uu);
    const float* cb = cdf + (size_t)b * F;
    const float target = uu[0] * cb[F - 1];
    int lo = 0, hi = F - 1;                              // first face whose cumulative area exceeds the target
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cb[mid] > target) hi = mid; else lo = mid + 1;
    }
    const float* vb = verts + (size_t)b * P * 3;
    const int ia = min(max(faces[lo * 3], 0), P - 1), ib = min(max(faces[lo * 3 + 1], 0), P - 1), ic = min(max(faces[lo * 3 + 2], 0), P - 1);
    const F3 a = ld3(vb + ia * 3), bb = ld3(vb + ib * 3), c = ld3(vb + ic * 3);
    const float r = sqrtf(uu[1]);
    const float w0 = 1.0f - r, w1 = r * (1.0f - uu[2]), w2 = r * uu[2];
    st3(points + ((size_t)b * n + i) * 3, w0 * a.x + w1 * bb.x + w2 * c.x, w0 * a.y + w1 * bb.y + w2 * c.y, w0 * a.z + w1 * bb.z + w2 * c.z);
    face_idx[(size_t)b * n + i] = lo;
    st3(bary + ((size_t)b * n + i) * 3, w0, w1, w2);
}

// grad_verts (zeroed by the caller) += barycentric weights x point gradients
__global__ __launch_bounds__(256) void mesh_sample_bwd_kernel(const int32_t* __restrict__ faces, const int32_t* __restrict__ face_idx,
                                                              const float* __restrict__ bary, const float* __restrict__ gpoints,
                                                              int P, int F, int n, float* __restrict__ gverts) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int f = min(max(face_idx[(size_t)b * n + i], 0), F - 1);
    const F3 w = ld3(bary + ((size_t)b * n + i) * 3), g = ld3(gpoints + ((size_t)b * n + i) * 3);
    float* gv = gverts + (size_t)b * P * 3;
    const float ws[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int vi = min(max(faces[f * 3 + k], 0), P - 1);
        atomicAdd(gv + vi * 3, ws[k] * g.x); atomicAdd(gv + vi * 3 + 1, ws[k] * g.y); atomicAdd(gv + vi * 3 + 2, ws[k] * g.z);
    }
}

}  // namespace vpn

using namespace vpn;

static int mesh_args(const void* verts, const void* faces, int B, int P, int F) {
    if (!verts || !faces) return VPN_E_BADARG;
    if (B <= 0 || P <= 0 || F <= 0) return VPN_E_BADARG;
    if (B > 65535 || (long long)B * P > 0x7fffffffLL / 4 || (long long)F > 0x7fffffffLL / 3) return VPN_E_TOOBIG;
    return 0;
}

extern "C" size_t vpn_mesh_raster_workspace(int B, int P) {
    if (B <= 0 || P <= 0) return 0;
    return (size_t)B * P * (sizeof(float4) + 2 * sizeof(float));     // projected vertices + their gradient
}

extern "C" int vpn_mesh_raster_fwd(const float* verts, const int32_t* faces, const float* cam, int B, int P, int F, int H, int W,
                                   float sigma, void* workspace, float* alpha, void* stream) {
    int rc = mesh_args(verts, faces, B, P, F);
    if (rc) return rc;
    if (!cam || !workspace || !alpha || H <= 0 || W <= 0 || !(sigma > 0.f) || ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float4* proj = (float4*)workspace;
    VPN_LAUNCH(mesh_project_kernel, dim3((P + 255) / 256, B), dim3(256), 0, s, verts, cam, P, proj);
    VPN_LAUNCH_CHECK();
    const int tiles_x = (W + R_TW - 1) / R_TW, tiles_y = (H + R_TH - 1) / R_TH;
    VPN_LAUNCH(mesh_raster_fwd_kernel, dim3((unsigned)tiles_x * tiles_y * B), dim3(64), 0, s, (const float4*)proj, faces, B, P, F, H, W,
               tiles_x, sigma, alpha);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_mesh_raster_bwd(const float* verts, const int32_t* faces, const float* cam, int B, int P, int F, int H, int W,
                                   float sigma, void* workspace, const float* alpha, const float* grad_alpha,
                                   float* grad_verts, void* stream) {
    int rc = mesh_args(verts, faces, B, P, F);
    if (rc) return rc;
    if (!cam || !workspace || !alpha || !grad_alpha || !grad_verts || H <= 0 || W <= 0 || !(sigma > 0.f) ||
        ((uintptr_t)workspace & 15) != 0) return VPN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    float4* proj = (float4*)workspace;                   // written by the forward call on the same workspace
    float* gproj = reinterpret_cast<float*>(proj + (size_t)B * P);
    hipError_t e = hipMemsetAsync(gproj, 0, (size_t)B * P * 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    const int tiles_x = (W + R_TW - 1) / R_TW, tiles_y = (H + R_TH - 1) / R_TH;
    VPN_LAUNCH(mesh_raster_bwd_kernel, dim3((unsigned)tiles_x * tiles_y * B), dim3(64), 0, s, (const float4*)proj, faces, B, P, F, H, W,
               tiles_x, sigma, alpha, grad_alpha, gproj);
    VPN_LAUNCH_CHECK();
    VPN_LAUNCH(mesh_project_bwd_kernel, dim3((P + 255) / 256, B), dim3(256), 0, s, verts, cam, P, (const float*)gproj, grad_verts);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_mesh_sample_fwd(const float* verts, const int32_t* faces, const float* u, uint64_t seed, uint64_t mesh_base,
                                   int B, int P, int F, int n, float* cdf, float* points, int32_t* face_idx, float* bary,
                                   void* stream) {
    int rc = mesh_args(verts, faces, B, P, F);
    if (rc) return rc;
    if (n <= 0 || !cdf || !points || !face_idx || !bary) return VPN_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    VPN_LAUNCH(mesh_cdf_kernel, dim3(B), dim3(1024), 0, s, verts, faces, P, F, cdf);
    VPN_LAUNCH_CHECK();
    VPN_LAUNCH(mesh_sample_fwd_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, verts, faces, (const float*)cdf, u, seed, mesh_base,
               P, F, n, points, face_idx, bary);
    VPN_LAUNCH_CHECK();
    return 0;
}

extern "C" int vpn_mesh_sample_bwd(const int32_t* faces, const int32_t* face_idx, const float* bary, const float* grad_points,
                                   int B, int P, int F, int n, float* grad_verts, void* stream) {
    if (!faces || !face_idx || !bary || !grad_points || !grad_verts) return VPN_E_BADARG;
    if (B <= 0 || P <= 0 || F <= 0 || n <= 0) return VPN_E_BADARG;
    if (B > 65535) return VPN_E_TOOBIG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_verts, 0, (size_t)B * P * 3 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    VPN_LAUNCH(mesh_sample_bwd_kernel, dim3((n + 255) / 256, B), dim3(256), 0, s, faces, face_idx, bary, grad_points, P, F, n, grad_verts);
    VPN_LAUNCH_CHECK();
    return 0;
}
