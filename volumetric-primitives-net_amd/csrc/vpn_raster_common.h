// vpn_raster_common.h — pieces of the primitive raster shared by raster.hip (tile kernels) and
// sampler.hip (the hot-path kernels that fold the raster's per-primitive work into the sampler's launches):
// camera, per-primitive records, the wave reduction and the finishing chain rule.  gfx950 only.
#pragma once
#include "vpn_common.h"

namespace vpn {

constexpr float R_TAN_HALF_FOV = 0.4571428511950223f;   // tan(49.13434207744484 deg / 2): kaolin v0.1 default fov
constexpr float R_X_CLAMP = 80.0f;
constexpr float R_E_CLAMP = 8.0f;
constexpr float R_EPS_H = 1e-3f;     // squareplus smoothing of relu(1 - m2) under the chord sqrt
constexpr float R_DELTA_S0 = 1e-12f;
constexpr float R_EPS_D = 1e-9f;
constexpr float R_X_CUT = 16.0f;     // primitives whose coverage logit is below -X_CUT on a tile are skipped: coverage < 1.2e-7
constexpr int R_TW = 16, R_TH = 16;  // pixel tile per wave
constexpr int R_PPL = 4;             // pixels per lane (row groups of 4 rows)
constexpr int R_CULL = 5;            // float4 per primitive staged in LDS for the tile mask: pixel box + conic (ellipsoid) or
                                     // + the silhouette hexagon of the inflated box (cuboid)
constexpr int R_FIN = 4 + R_CULL;    // first float4 of the finishing step's part of a record (camera basis and pose)
constexpr int R_REC = R_FIN + 7;     // float4 per primitive record in HBM: 4 ray coefficients, R_CULL culling, 7 finishing = 16

struct Camera {
    float eye[3], right[3], up[3], fwd[3];
    float dist;
};

// look-at camera of vertex_renderer.py:18 (set_look_at_parameters([azim],[elev],[dist]), degrees)
__device__ inline Camera make_camera(const float* cam) {
#pragma clang fp contract(off)            // same bits wherever it is inlined (see make_pose)
    Camera C;
    const float d = cam[0];
    const float el = cam[1] * 0.017453292519943295f, az = cam[2] * 0.017453292519943295f;
    float ce = cosf(el), se = sinf(el), ca = cosf(az), sa = sinf(az);
    C.eye[0] = d * ce * ca; C.eye[1] = d * se; C.eye[2] = d * ce * sa;
    float inv = 1.0f / sqrtf(C.eye[0] * C.eye[0] + C.eye[1] * C.eye[1] + C.eye[2] * C.eye[2]);
    float zx = C.eye[0] * inv, zy = C.eye[1] * inv, zz = C.eye[2] * inv;
    // right = normalize((0,1,0) x zax) ; up = zax x right
    float rx = zz, ry = 0.0f, rz = -zx;
    float rinv = 1.0f / sqrtf(rx * rx + rz * rz);
    rx *= rinv; rz *= rinv;
    C.right[0] = rx; C.right[1] = ry; C.right[2] = rz;
    C.up[0] = zy * rz - zz * ry; C.up[1] = zz * rx - zx * rz; C.up[2] = zx * ry - zy * rx;
    C.fwd[0] = -zx; C.fwd[1] = -zy; C.fwd[2] = -zz;
    C.dist = d;
    return C;
}

struct PrimGeo {   // per-primitive quantities that do not depend on the pixel
    float o[3], Mr[3], Mu[3], Mf[3];
};

__device__ inline void prim_geometry(const Camera& C, const Mat3& R, const float* v, const float* t, PrimGeo& G) {
#pragma clang fp contract(off)
    float e[3] = {C.eye[0] - t[0], C.eye[1] - t[1], C.eye[2] - t[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float iv = 1.0f / v[a];
        G.o[a] = (R.m[0][a] * e[0] + R.m[1][a] * e[1] + R.m[2][a] * e[2]) * iv;
        G.Mr[a] = (R.m[0][a] * C.right[0] + R.m[1][a] * C.right[1] + R.m[2][a] * C.right[2]) * iv;
        G.Mu[a] = (R.m[0][a] * C.up[0] + R.m[1][a] * C.up[1] + R.m[2][a] * C.up[2]) * iv;
        G.Mf[a] = (R.m[0][a] * C.fwd[0] + R.m[1][a] * C.fwd[1] + R.m[2][a] * C.fwd[2]) * iv;
    }
}

// Record of one primitive for one camera: out[0..3] = (o~|kind, Mr, Mu, Mf) so that d~ = Mf + px Mr + py Mu;
// out[4] = pixel bounding box of the culling region (jmin, jmax, imin, imax as int bits); out[5..8] = the region itself:
// ellipsoid: its conic in out[5..6]; cuboid: the silhouette hexagon of the inflated box, six half-planes
// a (x - cx) + b (y - cy) + 1 >= 0 as (a, b) pairs in out[5..7] and the centre (cx, cy) in out[8].
// ... from a camera and a pose that already exist (the sampler's forward launch: its pose lane has made the pose, the
// record lane the camera, side by side; both come from make_camera / make_pose, so the record is the same bit for bit)
__device__ inline void make_record_from(const Camera& C, const Pose& P, const float* __restrict__ prm, int kind, int H, int W,
                                        float sigma, float4 out[R_REC]);
// ... the same with every float4 handed to `put(i, value)` as soon as it exists (a caller short of registers stores it
// at once instead of holding all 14)
template <class Put>
__device__ inline void make_record_put(const Camera& C, const Pose& P, const float* __restrict__ prm, int kind, int H, int W,
                                       float sigma, Put&& put);

__device__ inline void make_record(const float* __restrict__ prm, int kind, const float* __restrict__ cam, int b, int H,
                                   int W, float sigma, float4 out[R_REC]) {
    const Camera C = make_camera(cam + b * 3);
    const Pose P = make_pose(prm[3], prm[4], prm[5], prm[6]);
    make_record_from(C, P, prm, kind, H, W, sigma, out);
}

__device__ inline void make_record_from(const Camera& C, const Pose& P, const float* __restrict__ prm, int kind, int H, int W,
                                        float sigma, float4 out[R_REC]) {
    make_record_put(C, P, prm, kind, H, W, sigma, [&](int i, const float4 v) { out[i] = v; });
}

template <class Put>
__device__ inline void make_record_put(const Camera& C, const Pose& P, const float* __restrict__ prm, int kind, int H, int W,
                                       float sigma, Put&& put) {
#pragma clang fp contract(off)            // the record does not depend on which kernel builds it
    float v[3] = {prm[0], prm[1], prm[2]};
    float t[3] = {prm[7], prm[8], prm[9]};
    PrimGeo G;
    prim_geometry(C, P.R, v, t, G);
    // out[7..13]: camera basis and pose as the finishing step of the backward needs them (it used to redo the six
    // sin / cos of make_camera and make_pose on one lane: ~1000 dependent instructions on its critical path)
    put(R_FIN + 0, make_float4(C.eye[0], C.eye[1], C.eye[2], C.right[0]));
    put(R_FIN + 1, make_float4(C.right[1], C.right[2], C.up[0], C.up[1]));
    put(R_FIN + 2, make_float4(C.up[2], C.fwd[0], C.fwd[1], C.fwd[2]));
    put(R_FIN + 3, make_float4(P.R.m[0][0], P.R.m[0][1], P.R.m[0][2], P.R.m[1][0]));
    put(R_FIN + 4, make_float4(P.R.m[1][1], P.R.m[1][2], P.R.m[2][0], P.R.m[2][1]));
    put(R_FIN + 5, make_float4(P.R.m[2][2], P.x, P.y, P.z));
    put(R_FIN + 6, make_float4(P.w, P.sh, P.ch, P.inv_len));
    put(0, make_float4(G.o[0], G.o[1], G.o[2], __int_as_float(kind)));
    put(1, make_float4(G.Mr[0], G.Mr[1], G.Mr[2], 0.f));
    put(2, make_float4(G.Mu[0], G.Mu[1], G.Mu[2], 0.f));
    put(3, make_float4(G.Mf[0], G.Mf[1], G.Mf[2], 0.f));
    // Culling region: rays whose squared miss distance m2 (scaled frame) is <= L2 = lam_cut^2, outside
    // of which the coverage logit (1 - m2)/sigma is below -X_CUT.  With d~ = M p, p = (px, py, 1) and
    // M = [Mr Mu Mf]:  m2 <= L2  <=>  q(p) = (u.p)^2 - c p^T G p >= 0,  u = M^T o~, G = M^T M,
    // c = |o~|^2 - L2: a conic in the image plane (an ellipse when the camera is outside the inflated
    // primitive).
    const float L2 = (1.0f + R_X_CUT * sigma) * 1.004f;
    const float txs = R_TAN_HALF_FOV * (float)W / (float)H;
    if (kind != VPN_SPHERE) {
        // Cuboid: lam <= lam_cut <=> the pixel's line meets the box inflated by lam_cut <=> the pixel lies in the
        // projection of that box, the convex hull of its 8 projected corners (all in front of the camera): a hexagon
        // (or quadrilateral) bounded by the box edges between a face that looks at the eye and one that does not.  (The
        // circumscribed sphere the ellipsoid path would use covers 1.5x the 8x8 quadrants this does at C3.)
        const float lamc = sqrtf(L2);
        float cc[3], ax[3][3];                       // box centre and inflated half-axis vectors in camera coordinates
        {
            const float w[3] = {t[0] - C.eye[0], t[1] - C.eye[1], t[2] - C.eye[2]};
            cc[0] = w[0] * C.right[0] + w[1] * C.right[1] + w[2] * C.right[2];
            cc[1] = w[0] * C.up[0] + w[1] * C.up[1] + w[2] * C.up[2];
            cc[2] = w[0] * C.fwd[0] + w[1] * C.fwd[1] + w[2] * C.fwd[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float h = lamc * v[i];
                ax[i][0] = (P.R.m[0][i] * C.right[0] + P.R.m[1][i] * C.right[1] + P.R.m[2][i] * C.right[2]) * h;
                ax[i][1] = (P.R.m[0][i] * C.up[0] + P.R.m[1][i] * C.up[1] + P.R.m[2][i] * C.up[2]) * h;
                ax[i][2] = (P.R.m[0][i] * C.fwd[0] + P.R.m[1][i] * C.fwd[1] + P.R.m[2][i] * C.fwd[2]) * h;
            }
        }
        // all 8 corners must be in front of the camera (and the eye outside the inflated box) for the hull argument
        const float reach = fabsf(ax[0][2]) + fabsf(ax[1][2]) + fabsf(ax[2][2]);
        const float fo[3] = {G.o[0] > lamc ? 1.0f : (G.o[0] < -lamc ? -1.0f : 0.0f), G.o[1] > lamc ? 1.0f : (G.o[1] < -lamc ? -1.0f : 0.0f),
                             G.o[2] > lamc ? 1.0f : (G.o[2] < -lamc ? -1.0f : 0.0f)};     // which face of each axis looks at the eye
        const bool ok = cc[2] - reach > 1.0e-3f * (fabsf(cc[2]) + reach) && (fo[0] != 0.0f || fo[1] != 0.0f || fo[2] != 0.0f);
        // every float4 leaves as soon as it exists (few live registers: this runs inside kernels with tight budgets)
        const float cx = ok ? cc[0] / cc[2] : 0.0f, cy = ok ? cc[1] / cc[2] : 0.0f;
        // silhouette edges: the axis pair (i, j) owns two of them unless neither axis has a face that looks at the eye
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2, k = 3 - i - j;
            const float fi = fo[i], fj = fo[j];
            float l[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};            // trivial half-planes
            if (ok && (fi != 0.0f || fj != 0.0f)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float si, sj;
                    if (e == 0) { si = fi != 0.0f ? fi : 1.0f; sj = fj != 0.0f ? (fi != 0.0f ? -fj : fj) : 1.0f; }
                    else { si = fi != 0.0f ? (fj != 0.0f ? -fi : fi) : -1.0f; sj = fj != 0.0f ? fj : -1.0f; }
                    float m[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) m[a] = cc[a] + si * ax[i][a] + sj * ax[j][a];
                    const float z0 = m[2] - ax[k][2], z1 = m[2] + ax[k][2];
                    const float x0 = (m[0] - ax[k][0]) / z0, y0 = (m[1] - ax[k][1]) / z0;
                    const float x1 = (m[0] + ax[k][0]) / z1, y1 = (m[1] + ax[k][1]) / z1;
                    const float la = y1 - y0, lb = x0 - x1;
                    const float vc = la * (cx - x0) + lb * (cy - y0);        // the projected centre lies strictly inside the hull
                    // a degenerate edge (seen end-on) keeps the trivial half-plane
                    if (fabsf(vc) > 1.0e-30f && fabsf(vc) < 3.0e38f) { l[e][0] = la / vc; l[e][1] = lb / vc; }
                }
            }
            put(5 + pr, make_float4(l[0][0], l[0][1], l[1][0], l[1][1]));
        }
        put(8, make_float4(cx, cy, 0.0f, 0.0f));
        int jmin = 0, jmax = W - 1, imin = 0, imax = H - 1;
        if (ok) {
            float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float s0 = (q & 1) ? 1.0f : -1.0f, s1 = (q & 2) ? 1.0f : -1.0f, s2 = (q & 4) ? 1.0f : -1.0f;
                const float z = cc[2] + s0 * ax[0][2] + s1 * ax[1][2] + s2 * ax[2][2];
                const float x = (cc[0] + s0 * ax[0][0] + s1 * ax[1][0] + s2 * ax[2][0]) / z;
                const float y = (cc[1] + s0 * ax[0][1] + s1 * ax[1][1] + s2 * ax[2][1]) / z;
                xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); ylo = fminf(ylo, y); yhi = fmaxf(yhi, y);
            }
            const float jl = (xlo / txs + 1.0f) * (0.5f * W) - 0.5f, jh = (xhi / txs + 1.0f) * (0.5f * W) - 0.5f;
            const float il = (1.0f - yhi / R_TAN_HALF_FOV) * (0.5f * H) - 0.5f, ih = (1.0f - ylo / R_TAN_HALF_FOV) * (0.5f * H) - 0.5f;
            jmin = (int)fminf(fmaxf(floorf(jl) - 1.0f, -1.0e6f), 1.0e6f);
            jmax = (int)fminf(fmaxf(ceilf(jh) + 1.0f, -1.0e6f), 1.0e6f);
            imin = (int)fminf(fmaxf(floorf(il) - 1.0f, -1.0e6f), 1.0e6f);
            imax = (int)fminf(fmaxf(ceilf(ih) + 1.0f, -1.0e6f), 1.0e6f);
        }
        put(4, make_float4(__int_as_float(jmin), __int_as_float(jmax), __int_as_float(imin), __int_as_float(imax)));
        return;
    }
    const float* col[3] = {G.Mr, G.Mu, G.Mf};
    float u[3], Gm[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        u[i] = col[i][0] * G.o[0] + col[i][1] * G.o[1] + col[i][2] * G.o[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) Gm[i][j] = col[i][0] * col[j][0] + col[i][1] * col[j][1] + col[i][2] * col[j][2];
    }
    const float c = (G.o[0] * G.o[0] + G.o[1] * G.o[1] + G.o[2] * G.o[2]) - L2;
    float Q[3][3], qmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { Q[i][j] = u[i] * u[j] - c * Gm[i][j]; qmax = fmaxf(qmax, fabsf(Q[i][j])); }
    const float qs = 1.0f / qmax;                       // positive scaling keeps the sign of q
    const float A00 = Q[0][0] * qs, A01 = Q[0][1] * qs, A11 = Q[1][1] * qs;
    const float b0 = Q[0][2] * qs, b1 = Q[1][2] * qs, c0 = Q[2][2] * qs;
    const float det = A00 * A11 - A01 * A01;
    int jmin = 0, jmax = W - 1, imin = 0, imax = H - 1;
    float valid = 0.0f;
    if (c > 0.0f && A00 < 0.0f && det > 1e-12f) {       // proper ellipse; anything else: keep the full image
        const float xs = -(A11 * b0 - A01 * b1) / det, ys = -(A00 * b1 - A01 * b0) / det;
        const float qstar = c0 + b0 * xs + b1 * ys;     // value at the centre
        if (qstar > 0.0f) {
            const float hx = sqrtf(qstar * (-A11) / det), hy = sqrtf(qstar * (-A00) / det);
            const float jl = ((xs - hx) / txs + 1.0f) * (0.5f * W) - 0.5f, jh = ((xs + hx) / txs + 1.0f) * (0.5f * W) - 0.5f;
            const float il = (1.0f - (ys + hy) / R_TAN_HALF_FOV) * (0.5f * H) - 0.5f;
            const float ih = (1.0f - (ys - hy) / R_TAN_HALF_FOV) * (0.5f * H) - 0.5f;
            jmin = (int)fminf(fmaxf(floorf(jl) - 1.0f, -1.0e6f), 1.0e6f);
            jmax = (int)fminf(fmaxf(ceilf(jh) + 1.0f, -1.0e6f), 1.0e6f);
            imin = (int)fminf(fmaxf(floorf(il) - 1.0f, -1.0e6f), 1.0e6f);
            imax = (int)fminf(fmaxf(ceilf(ih) + 1.0f, -1.0e6f), 1.0e6f);
            valid = 1.0f;
        } else if (qstar < 0.0f) {                       // empty region: never visible
            jmin = 1; jmax = 0; imin = 1; imax = 0;
            valid = 1.0f;
        }
    }
    put(4, make_float4(__int_as_float(jmin), __int_as_float(jmax), __int_as_float(imin), __int_as_float(imax)));
    put(5, make_float4(A00, A01, A11, valid));
    put(6, make_float4(b0, b1, c0, det));
    put(7, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    put(8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
}

// Transposing butterfly: 16 per-lane values -> lane L holds the wave total of value (L >> 2).
// wave_reduce16_swap: same pairs and the same additions as the shuffle form below (a level sends half of the values to the partner lane
// and keeps the other half: bit-identical sums), but without the LDS crossbar: gfx950's half exchanges
// (v_permlane32_swap / v_permlane16_swap: ONE instruction does the send and the receive of a pair of values) for the
// partners 32 and 16 lanes away, DPP row rotations and quad permutations inside the 16-lane rows.  The shuffle form was 17
// ds_bpermute + 30 selects per call, six LDS round trips deep, once per (tile, primitive) in the tile kernels.
// Two forms: the tile kernels are bound by VALU issue, where the shuffles' LDS-pipe instructions are free and the swaps are
// not (raster 45.0 us with shuffles, 46.8 with swaps); the fused backward is a chain of latencies, where it is the other way
// round (17.4 -> 16.7 us).
__device__ inline float wave_reduce16_swap(float v[16]) {
    const int lane = threadIdx.x & 63;
    auto swap32 = [](float& a, float& b) {     // lanes 32-63 of a <-> lanes 0-31 of b
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
    auto swap16 = [](float& a, float& b) {     // odd 16-lane rows of a <-> even rows of b
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
#define VPN_DPP(x, ctrl, bank) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, bank, false))
#pragma unroll
    for (int i = 0; i < 8; ++i) { float a = v[i], b = v[i + 8]; swap32(a, b); v[i] = a + b; }     // lower half: values 0-7, upper: 8-15
#pragma unroll
    for (int i = 0; i < 4; ++i) { float a = v[i], b = v[i + 4]; swap16(a, b); v[i] = a + b; }     // even rows: i, odd rows: i + 4
    {   // partner 8 lanes away inside the row (a rotation by 8 is its own inverse)
        const bool hi = lane & 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float a = v[i] + VPN_DPP(v[i], 0x128, 0xf), b = v[i + 2] + VPN_DPP(v[i + 2], 0x128, 0xf);
            v[i] = hi ? b : a;
        }
    }
    float r;
    {   // partner 4 lanes away: a rotation by 4 for one half of the 4-lane banks, by 12 for the other
        auto x4 = [](float x) {
            int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xa, false);
            t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x12c, 0xf, 0x5, false);
            return __int_as_float(t);
        };
        const float a = v[0] + x4(v[0]), b = v[1] + x4(v[1]);
        r = (lane & 4) ? b : a;
    }
    r += VPN_DPP(r, 0x4e, 0xf);                // quad_perm [2,3,0,1]: partner 2 lanes away
    r += VPN_DPP(r, 0xb1, 0xf);                // quad_perm [1,0,3,2]: partner 1 lane away
#undef VPN_DPP
    return r;
}
__device__ inline float wave_reduce16(float v[16]) {
    const int lane = threadIdx.x & 63;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float send = hi ? v[i] : v[i + 8], keep = hi ? v[i + 8] : v[i];
            v[i] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float send = hi ? v[i] : v[i + 4], keep = hi ? v[i + 4] : v[i];
            v[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool hi = lane & 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float send = hi ? v[i] : v[i + 2], keep = hi ? v[i + 2] : v[i];
            v[i] = keep + __shfl_xor(send, 8, 64);
        }
    }
    float r;
    {
        const bool hi = lane & 4;
        float send = hi ? v[0] : v[1], keep = hi ? v[1] : v[0];
        r = keep + __shfl_xor(send, 4, 64);
    }
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// does the region q >= 0 of the conic touch the rectangle [x0,x1] x [y0,y1] (slope units)?  Exact for an
// ellipse: the centre if it is inside, otherwise the maximum of the concave quadratic over the 4 edges.
__device__ inline bool conic_hits_rect(const float4 qa, const float4 qb, float x0, float x1, float y0, float y1) {
    const float A00 = qa.x, A01 = qa.y, A11 = qa.z, b0 = qb.x, b1 = qb.y, c0 = qb.z, det = qb.w;
    const float idet = __builtin_amdgcn_rcpf(det);
    const float xs = -(A11 * b0 - A01 * b1) * idet, ys = -(A00 * b1 - A01 * b0) * idet;
    if (xs >= x0 && xs <= x1 && ys >= y0 && ys <= y1) return true;
    float best = -1.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float x = e ? x1 : x0;                                         // vertical edges
        const float lin = A01 * x + b1, cst = (A00 * x + 2.0f * b0) * x + c0;
        const float yv = fminf(fmaxf(-lin * __builtin_amdgcn_rcpf(A11), y0), y1);
        best = fmaxf(best, (A11 * yv + 2.0f * lin) * yv + cst);
        const float y = e ? y1 : y0;                                         // horizontal edges
        const float lin2 = A01 * y + b0, cst2 = (A11 * y + 2.0f * b1) * y + c0;
        const float xv = fminf(fmaxf(-lin2 * __builtin_amdgcn_rcpf(A00), x0), x1);
        best = fmaxf(best, (A00 * xv + 2.0f * lin2) * xv + cst2);
    }
    return best >= 0.0f;
}

// does the convex region  a_e (x - cx) + b_e (y - cy) + 1 >= 0, e < 6  (lines in l0, l1, l2 as (a, b) pairs, centre in ctr.xy)
// touch the rectangle?  Separating axes of a convex polygon and a rectangle: the polygon's edges (here) and the
// rectangle's (the pixel-box test of the caller).  A trivial line (0, 0) never rejects.
__device__ inline bool hexagon_hits_rect(const float4 l0, const float4 l1, const float4 l2, const float4 ctr, float x0, float x1,
                                         float y0, float y1) {
    const float dx0 = x0 - ctr.x, dx1 = x1 - ctr.x, dy0 = y0 - ctr.y, dy1 = y1 - ctr.y;
    auto out = [&](float a, float b) { return fmaxf(a * dx0, a * dx1) + fmaxf(b * dy0, b * dy1) + 1.0f < 0.0f; };   // the whole rectangle outside
    return !(out(l0.x, l0.y) || out(l0.z, l0.w) || out(l1.x, l1.y) || out(l1.z, l1.w) || out(l2.x, l2.y) || out(l2.z, l2.w));
}

// does the primitive whose culling record is cr[0 .. R_CULL) (pixel box, then conic or hexagon) touch the 16x16 tile at (c0, r0)?
__device__ inline bool prim_hits_tile(int kind, const float4* cr, int c0, int r0, int H, int W,
                                      int R_TW = vpn::R_TW, int R_TH = vpn::R_TH) {       // also used for the 8x8 quadrants
    const float4 bb = cr[0];
    const int jmin = __float_as_int(bb.x), jmax = __float_as_int(bb.y);
    const int imin = __float_as_int(bb.z), imax = __float_as_int(bb.w);
    bool vis = (jmin <= c0 + R_TW - 1) && (jmax >= c0) && (imin <= r0 + R_TH - 1) && (imax >= r0);
    if (vis) {
        // tile rectangle in slope units, half a pixel of margin on every side
        const float sx = 2.0f * (R_TAN_HALF_FOV * (float)W / (float)H) / (float)W, sy = 2.0f * R_TAN_HALF_FOV / (float)H;
        const float x0 = ((float)c0 - 0.5f * (float)W) * sx, x1 = ((float)(c0 + R_TW) - 0.5f * (float)W) * sx;
        const float y1 = (0.5f * (float)H - (float)r0) * sy, y0 = (0.5f * (float)H - (float)(r0 + R_TH)) * sy;
        if (kind == VPN_SPHERE) {
            const float4 qa = cr[1];
            if (qa.w != 0.0f) vis = conic_hits_rect(qa, cr[2], x0, x1, y0, y1);
        } else {
            vis = hexagon_hits_rect(cr[1], cr[2], cr[3], cr[4], x0, x1, y0, y1);
        }
    }
    return vis;
}


// ---- launch order of the training step's tile waves.  One workgroup per image: every tile is tested against the K
// primitives of the image (the test of the tile kernels; two lanes share a tile, odd / even primitives), the masks go
// where the tile kernels expect them, and the tiles are sorted by their number of visible primitives, heaviest first
// (counting sort in LDS).  Runs as a rider in the tail of the Chamfer scan's launch (chamfer.hip), between the launch
// that writes the records and the one that reads masks and order.
// One entry per (image, launch rank) of the training step's tile waves: which tile, its visible primitives, and which of
// them reach each of its four 8x8 quadrants -- everything a tile wave used to find out for itself (order -> mask ->
// staged culling records -> quadrant test) in ONE 48-byte load.
struct TileEntry {
    unsigned int tile, n;                     // tile index inside the image; number of visible primitives
    unsigned long long mask;                  // bit k <=> primitive k may touch the tile (K <= 64)
    unsigned long long q[4];                  // four bits per STAGED SLOT (slot j = the j-th visible primitive; bit qd = column half
                                              // + 2 * row half of the quadrant), sixteen slots per word: the words quadrant_bits makes
};
static_assert(sizeof(TileEntry) == 48, "three 16-byte loads");

struct RasterOrderJob {
    const float4* rec = nullptr;              // records [B][K][R_REC] (vpn_hotpath_sample_fwd)
    unsigned long long* masks = nullptr;      // out: [B][ntile] (K <= 64: one word per tile), read by the finishing step
    TileEntry* entries = nullptr;             // out: [B][ntile] by launch rank (heaviest tile first), then [B][ntile] by tile (scratch)
    int B = 0, K = 0, H = 0, W = 0, tiles_x = 0, ntile = 0;
};

// scratch (LDS): K * R_CULL float4 of cull records + K kinds + (K + 2) ints + ntile bytes; R_ORDER_MAX_TILES bounds the last term
constexpr int R_ORDER_MAX_TILES = 16384;      // 2048 x 2048 pixels
constexpr int R_ORDER_MAX_PRIMS = 64;         // one mask word per tile
__host__ __device__ inline size_t raster_order_scratch(int K, int ntile) {
    return (size_t)K * R_CULL * 16 + (size_t)K * 4 + (size_t)(K + 2) * 4 + (size_t)ntile;
}
template <int THREADS>
__device__ __forceinline__ void raster_order_wg(const RasterOrderJob& J, int b, void* scratch) {
    float4* cull = reinterpret_cast<float4*>(scratch);
    int* knd = reinterpret_cast<int*>(cull + R_CULL * J.K);      // kind of every primitive
    int* hist = knd + J.K;                                       // hist[c] -> start of the bucket of popcount c (descending)
    unsigned char* pops = reinterpret_cast<unsigned char*>(hist + J.K + 2);   // visible primitives per tile (K <= 64)
    const float4* rec_b = J.rec + (size_t)b * J.K * R_REC;
    for (int i = threadIdx.x; i < R_CULL * J.K; i += THREADS) cull[i] = rec_b[(size_t)(i / R_CULL) * R_REC + 4 + i % R_CULL];
    for (int k = threadIdx.x; k < J.K; k += THREADS) knd[k] = __float_as_int(rec_b[(size_t)k * R_REC].w);
    for (int i = threadIdx.x; i <= J.K + 1; i += THREADS) hist[i] = 0;
    __syncthreads();
    unsigned long long* mrow = J.masks + (size_t)b * J.ntile;
    TileEntry* by_rank = J.entries + (size_t)b * J.ntile;
    TileEntry* by_tile = J.entries + ((size_t)J.B + b) * J.ntile;
    auto both = [](unsigned long long v) {                                    // OR over the two lanes of a tile
        return v | ((unsigned long long)__shfl_xor((unsigned)(v >> 32), 1, 64) << 32) | (unsigned)__shfl_xor((unsigned)v, 1, 64);
    };
    // pass 1: masks (tile level, then the four quadrants of the visible pairs) and the histogram of the popcounts
    for (int it = threadIdx.x; it < 2 * J.ntile; it += THREADS) {            // pairs of lanes: THREADS and it are even together
        const int tile = it >> 1, par = it & 1;
        const int ty = tile / J.tiles_x, tx = tile - ty * J.tiles_x;
        unsigned long long m = 0ull, q[4] = {0ull, 0ull, 0ull, 0ull};
        for (int k = par; k < J.K; k += 2)
            if (prim_hits_tile(knd[k], cull + R_CULL * k, tx * R_TW, ty * R_TH, J.H, J.W)) m |= 1ull << k;
        unsigned long long own = m;                                          // this lane's visible primitives (its parity)
        m = both(m);
        // quadrants of the visible pairs only (a handful per lane): slot j = rank of primitive k among the tile's visible
        // ones = the slot the tile wave stages it in; four bits per slot, sixteen slots per word
        for (; own; own &= own - 1ull) {
            const int k = __builtin_ctzll(own), j = __builtin_popcountll(m & ((1ull << k) - 1ull));
            unsigned nib = 0u;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                if (prim_hits_tile(knd[k], cull + R_CULL * k, tx * R_TW + 8 * (qd & 1), ty * R_TH + 8 * (qd >> 1), J.H, J.W, 8, 8)) nib |= 1u << qd;
            const unsigned long long sh = (unsigned long long)nib << (4 * (j & 15));
            q[0] |= (j >> 4) == 0 ? sh : 0ull; q[1] |= (j >> 4) == 1 ? sh : 0ull; q[2] |= (j >> 4) == 2 ? sh : 0ull; q[3] |= (j >> 4) == 3 ? sh : 0ull;
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) q[qd] = both(q[qd]);
        if (par == 0) {
            const int pop = __builtin_popcountll(m);
            mrow[tile] = m;
            TileEntry e;
            e.tile = (unsigned)tile; e.n = (unsigned)pop; e.mask = m;
            e.q[0] = q[0]; e.q[1] = q[1]; e.q[2] = q[2]; e.q[3] = q[3];
            by_tile[tile] = e;
            pops[tile] = (unsigned char)pop;
            atomicAdd(&hist[J.K - pop], 1);                                    // bucket 0 = all K primitives visible
        }
    }
    __syncthreads();                                                          // also orders this workgroup's by_tile writes before its reads below
    if (threadIdx.x == 0) {                                                   // exclusive prefix over K + 1 buckets
        int run = 0;
        for (int c = 0; c <= J.K; ++c) { const int n = hist[c]; hist[c] = run; run += n; }
    }
    __syncthreads();
    // pass 2: every tile takes the next place of its bucket (the order inside a bucket is whatever the atomics give:
    // it schedules, it does not change a result)
    const uint4* src = reinterpret_cast<const uint4*>(by_tile);
    uint4* dst = reinterpret_cast<uint4*>(by_rank);
    for (int tile = threadIdx.x; tile < J.ntile; tile += THREADS) {
        const int pos = atomicAdd(&hist[J.K - (int)pops[tile]], 1);
        const uint4 e0 = src[3 * tile], e1 = src[3 * tile + 1], e2 = src[3 * tile + 2];
        dst[3 * pos] = e0; dst[3 * pos + 1] = e1; dst[3 * pos + 2] = e2;
    }
}

struct RasterPrep {       // what a kernel outside raster.hip needs to write the raster records (rec == nullptr: nothing to do)
    const float* cam = nullptr;
    float4* rec = nullptr;
    int* zero_me = nullptr;            // head of the loss workspace: {arrival counter, spare, effective Philox seed (u64)},
                                       // then per sample a float4 (sil sum, depth sum, cd, arrival counter of its tiles)
    int H = 0, W = 0;
    float sigma = 0.f;
};

struct RasterFinish {     // what a kernel outside raster.hip needs to run the finishing step (partial == nullptr: nothing to do)
    const float4* rec = nullptr;       // the records (camera basis and pose live in float4 7..13 of each)
    const unsigned long long* masks = nullptr;
    const float* partial = nullptr;
    const float* scale = nullptr;      // device scalar: upstream gradient of the fused total (nullptr = 1)
    int ntile = 0, words = 0;
};

// Finishing step of the raster backward for primitive bk = b * K + k.
// (1) gather: lane sums the partials of the tiles  first + lane + stride * i  whose mask holds k (fixed order);
__device__ inline void raster_finish_gather(int bk, int K, int ntile, int words, const unsigned long long* __restrict__ masks,
                                            const float* __restrict__ partial, int first, int stride, float v[16]) {
    const int lane = threadIdx.x & 63;
    const int b = bk / K, k = bk - b * K;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.0f;
    // 4 tiles per lane and round: the 4 mask words are loaded together, then the (up to) 12 float4 of the pairs
    // that exist -- two memory round trips per round instead of eight
    for (int t0 = first; t0 < ntile; t0 += 4 * stride) {
        bool has[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tile = t0 + u * stride + lane;
            const unsigned long long m = (tile < ntile && t0 + u * stride < ntile) ? masks[((size_t)b * ntile + tile) * words + (k >> 6)] : 0ull;
            has[u] = (m >> (k & 63)) & 1ull;                                       // nothing was written for the other pairs
        }
        float4 pa[4], pc[4], pd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4* src = reinterpret_cast<const float4*>(partial + ((size_t)bk * ntile + (t0 + u * stride + lane)) * 12);
            pa[u] = pc[u] = pd[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has[u]) { pa[u] = src[0]; pc[u] = src[1]; pd[u] = src[2]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!has[u]) continue;
            v[0] += pa[u].x; v[1] += pa[u].y; v[2] += pa[u].z; v[3] += pa[u].w;
            v[4] += pc[u].x; v[5] += pc[u].y; v[6] += pc[u].z; v[7] += pc[u].w;
            v[8] += pd[u].x; v[9] += pd[u].y; v[10] += pd[u].z; v[11] += pd[u].w;
        }
    }
}

// (2) chain rule from the 12 summed ray-coefficient gradients G to r[10] = d loss / d(v0 v1 v2 q0 q1 q2 q3 t0 t1 t2)
// (one thread, upstream gradient 1).
__device__ inline void raster_finish_chain(const float* __restrict__ params, const float4* __restrict__ rec, int bk, int K,
                                           const float G[12], float r[10]) {
    const float* prm = params + (size_t)bk * VPN_PARAM_STRIDE;
    const float4* rk = rec + (size_t)bk * R_REC + R_FIN;              // camera basis and pose saved by make_record
    const float4 c0 = rk[0], c1 = rk[1], c2 = rk[2], p0 = rk[3], p1 = rk[4], p2 = rk[5], p3 = rk[6];
    Camera C;
    C.eye[0] = c0.x; C.eye[1] = c0.y; C.eye[2] = c0.z; C.right[0] = c0.w; C.right[1] = c1.x; C.right[2] = c1.y;
    C.up[0] = c1.z; C.up[1] = c1.w; C.up[2] = c2.x; C.fwd[0] = c2.y; C.fwd[1] = c2.z; C.fwd[2] = c2.w; C.dist = 0.0f;
    Pose P;
    P.R.m[0][0] = p0.x; P.R.m[0][1] = p0.y; P.R.m[0][2] = p0.z; P.R.m[1][0] = p0.w; P.R.m[1][1] = p1.x; P.R.m[1][2] = p1.y;
    P.R.m[2][0] = p1.z; P.R.m[2][1] = p1.w; P.R.m[2][2] = p2.x; P.x = p2.y; P.y = p2.z; P.z = p2.w;
    P.w = p3.x; P.sh = p3.y; P.ch = p3.z; P.inv_len = p3.w;
    const float vv[3] = {prm[0], prm[1], prm[2]};
    const float t[3] = {prm[7], prm[8], prm[9]};
    PrimGeo Ge;
    prim_geometry(C, P.R, vv, t, Ge);
    const float e[3] = {C.eye[0] - t[0], C.eye[1] - t[1], C.eye[2] - t[2]};
    float gv[3], gyo[3], gyr[3], gyu[3], gyf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float iv = 1.0f / vv[a];
        gv[a] = -(G[a] * Ge.o[a] + G[3 + a] * Ge.Mr[a] + G[6 + a] * Ge.Mu[a] + G[9 + a] * Ge.Mf[a]) * iv;
        gyo[a] = G[a] * iv; gyr[a] = G[3 + a] * iv; gyu[a] = G[6 + a] * iv; gyf[a] = G[9 + a] * iv;
    }
    float gR[3][3], gt[3], gq[4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
            gR[r][a] = e[r] * gyo[a] + C.right[r] * gyr[a] + C.up[r] * gyu[a] + C.fwd[r] * gyf[a];
        gt[r] = -(P.R.m[r][0] * gyo[0] + P.R.m[r][1] * gyo[1] + P.R.m[r][2] * gyo[2]);
    }
    pose_backward(P, prm[3], prm[4], prm[5], gR, gq);
    r[0] = gv[0]; r[1] = gv[1]; r[2] = gv[2]; r[3] = gq[0]; r[4] = gq[1]; r[5] = gq[2]; r[6] = gq[3];
    r[7] = gt[0]; r[8] = gt[1]; r[9] = gt[2];
}

// both steps by ONE WAVE (64 consecutive tiles per round); lane 0 returns r
__device__ inline void raster_finish_wave(const float* __restrict__ params, const float4* __restrict__ rec, int bk, int K,
                                          int ntile, int words, const unsigned long long* __restrict__ masks,
                                          const float* __restrict__ partial, float r[10]) {
    const int lane = threadIdx.x & 63;
    float v[16];
    raster_finish_gather(bk, K, ntile, words, masks, partial, 0, 64, v);
    const float tot = wave_reduce16(v);
    float G[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) G[i] = __shfl(tot, i * 4, 64);
    if (lane != 0) return;
    raster_finish_chain(params, rec, bk, K, G, r);
}

}  // namespace vpn
