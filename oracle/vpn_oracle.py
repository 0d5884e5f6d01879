"""CPU oracle for the volumetric-primitive hot path.  TEST INFRASTRUCTURE ONLY.

This file is a pure-PyTorch (CPU) restatement of the reference algorithm for the
sampler / transform / Chamfer / VPDiverse path, plus the specification of the
new primitive soft raster.  It is the checker the HIP kernels are compared
against.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; nothing under `volumetric-primitives-net_amd/` does.

Pinning (see DESIGN.md "Oracle"):
  * sampler / transform / Chamfer / VPDiverse / view<->obj: PINNED.  The
    functions below are checked against golden vectors produced by importing the
    reference's own pure-PyTorch leaf files (`oracle/make_golden.py`, fixtures in
    `tests/golden/*.npz`, test `tests/test_oracle_golden.py`).
  * soft raster (`raster_*`): PARITY UNPINNED.  The reference renders through
    kaolin's DIBRenderer (modules/render/vertex_renderer.py:2-7,24), which is not
    in the reference tree, not pinned to a version and not installable here.  The
    raster below is the specification of a new operator behind the reference's
    render surface; the HIP kernel is compared against this specification.

Every function cites the reference file:line (relative to the reference root) it
restates.  All functions are dtype-generic: fp32 is the parity dtype, fp64 is
used by tests to bound the fp32 rounding noise of both implementations.
"""
import math
import torch

# modules/sampling/sphere.py:7 and modules/transform/rotate.py:4 (fp32 value of pi)
PI = 3.1415927410125732

SPHERE = 0   # ellipsoid primitive (reference: Sampling.sphere_sampling)
CUBOID = 1   # box primitive      (reference: Sampling.cuboid_sampling)


# --------------------------------------------------------------------------
# pose: axis-angle "quaternion" -> rotation matrix      (modules/transform/rotate.py)
# --------------------------------------------------------------------------
def refine_quaternions(q):
    """rotate.py:59-72.  q (B,4) = (axis xyz, angle in turns).  The axis is NOT
    pre-normalised; the 4-vector (axis*sin(h), cos(h)) is normalised instead."""
    h = torch.div((torch.remainder(q[:, 3], 1) * 2) * PI, 2)          # rotate.py:63
    s = torch.sin(h)
    r = torch.cat([q[:, :3] * s[:, None], torch.cos(h)[:, None]], 1)   # rotate.py:66-67
    length = torch.sqrt((r * r).sum(1))                                # rotate.py:69 (torch.norm)
    return r / length[:, None]                                         # rotate.py:70


def rotation_matrices(q):
    """rotate.py:28-46 built functionally (the reference's in-place writes into a
    leaf fail on CPU: SURVEY.md finding 3).  Returns (B,3,3)."""
    r = refine_quaternions(q)
    x, y, z, w = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    row0 = torch.stack([x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw)], 1)
    row1 = torch.stack([2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw)], 1)
    row2 = torch.stack([2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2], 1)
    return torch.stack([row0, row1, row2], 1)


def rotate_points(points, q):
    """rotate.py:7-25.  points (B,N,3), q (B,4) -> (B,N,3) = (R p^T)^T."""
    R = rotation_matrices(q)
    return torch.bmm(R, points.permute(0, 2, 1)).permute(0, 2, 1)


def translate_points(points, t):
    """translate.py:4-8."""
    return points + t[:, None, :]


def transform_points(points, q, t):
    """transform.py:6-9."""
    return translate_points(rotate_points(points, q), t)


# --------------------------------------------------------------------------
# sampler                                                 (modules/sampling/*)
# --------------------------------------------------------------------------
def sphere_canonical(u1, u2):
    """sphere.py:22-43 without the scaling.  u1,u2 (B,N) uniform draws (elev draw
    first, azim draw second: sphere.py:26-27).  Returns unit points (B,N,3)."""
    elev = -torch.acos(1 - 2 * u1) + PI * 0.5          # sphere.py:26
    azim = u2 * 2 * PI                                 # sphere.py:27
    ce = torch.cos(elev)
    return torch.stack([ce * torch.sin(azim), torch.sin(elev), ce * torch.cos(azim)], 2)  # :38-40


def sphere_sampling(v, q, t, u1, u2):
    """Sampling.sphere_sampling (sampling.py:25-37) with explicit uniform draws."""
    pts = sphere_canonical(u1, u2) * v[:, None, :]     # sphere.py:31-33
    return transform_points(pts, q, t)


def cuboid_face_counts(v, n):
    """cuboid.py:30-53.  v (B,3) -> int32 (B,6); faces +w,-w,+h,-h,+d,-d; the
    first five are round-half-even(n*area/total), the remainder goes to face 5."""
    w, h, d = v[:, 0:1], v[:, 1:2], v[:, 2:3]
    hd, dw, wh = h * d, d * w, w * h
    area = torch.cat([hd, hd, dw, dw, wh, wh], 1)
    total = (hd + dw + wh) * 2
    weight = area / total
    cnt = (torch.full_like(weight, n) * weight).round().int()
    cnt[:, 5] = n - cnt[:, :5].sum(1)
    return cnt


def cuboid_canonical(u, counts):
    """cuboid.py:56-101 expressed as a coefficient c with p_c = c * (w,h,d).
    u (B,N,3) uniform draws; counts (B,6).  Free coordinates are (2u-1); the
    points in index range [sum_{g<f} n_g, sum_{g<=f} n_g) get coordinate f//2
    overwritten with +-1 (cuboid.py:88-99: by index range, not by draw)."""
    B, N, _ = u.shape
    c = -1 + 2 * u                                      # cuboid.py:66
    start = torch.zeros(B, dtype=torch.long)
    idx = torch.arange(N)[None, :]
    c = c.clone()
    for f in range(6):
        num = counts[:, f].long()
        sel = (idx >= start[:, None]) & (idx < (start + num)[:, None])   # cuboid.py:95-97
        sign = -1.0 if f % 2 else 1.0
        cf = c[:, :, f // 2]
        c[:, :, f // 2] = torch.where(sel, torch.full_like(cf, sign), cf)
        start = start + num
    return c


def cuboid_sampling(v, q, t, u):
    """Sampling.cuboid_sampling (sampling.py:11-23) with explicit uniform draws.
    The face counts are integers: no gradient flows through them (the reference's
    `.round().int()` cuts the graph, cuboid.py:46)."""
    counts = cuboid_face_counts(v.detach(), u.shape[1])
    c = cuboid_canonical(u, counts)
    return transform_points(c * v[:, None, :], q, t)


def sample_primitives(params, types, u):
    """train.py:105-120 (sample_predict_points) on a packed tensor.
    params (B,K,10) = (v3,q4,t3); types list[K] of SPHERE/CUBOID; u (B,K,n,3)
    uniform draws (spheres use u[...,0] for elev and u[...,1] for azim).
    Returns (B, K*n, 3), primitive-major like torch.cat(dim=1) at train.py:119."""
    B, K, _ = params.shape
    out = []
    for k in range(K):
        v, q, t = params[:, k, 0:3], params[:, k, 3:7], params[:, k, 7:10]
        if types[k] == SPHERE:
            out.append(sphere_sampling(v, q, t, u[:, k, :, 0], u[:, k, :, 1]))
        else:
            out.append(cuboid_sampling(v, q, t, u[:, k]))
    return torch.cat(out, 1)


# --------------------------------------------------------------------------
# view <-> object transforms                     (modules/transform/transform.py)
# --------------------------------------------------------------------------
def _axis_q(axis, angle):
    B = angle.shape[0]
    a = torch.tensor([axis], dtype=angle.dtype).repeat(B, 1)
    return torch.cat([a, angle.view(-1, 1)], 1)


def rotate_points_forward_x_axis(points, angles):
    """transform.py:76-94.  angles in degrees."""
    return rotate_points(points, _axis_q([1.0, 0.0, 0.0], angles.view(-1) / 360))


def obj_to_view_points(points, dists, elevs, azims):
    """transform.py:50-73."""
    elevs, azims = elevs.view(-1, 1) / 360, azims.view(-1, 1) / 360
    B = points.shape[0]
    y = torch.tensor([[0.0, 1.0, 0.0]], dtype=points.dtype).repeat(B, 1)
    q = _axis_q([0.0, 0.0, -1.0], elevs)
    points = rotate_points(points, q)
    y = rotate_points(y[:, None, :], q)[:, 0]
    points = rotate_points(points, torch.cat([y, azims], 1))
    return points / dists.view(-1, 1, 1)


def view_to_obj_points(points, dists, elevs, azims, angles):
    """transform.py:21-47."""
    elevs, azims = elevs.view(-1, 1) / 360, azims.view(-1, 1) / 360
    points = rotate_points_forward_x_axis(points, -angles)
    B = points.shape[0]
    y = torch.tensor([[0.0, 1.0, 0.0]], dtype=points.dtype).repeat(B, 1)
    y = rotate_points(y[:, None, :], _axis_q([0.0, 0.0, -1.0], elevs))[:, 0]
    points = rotate_points(points, torch.cat([y, -azims], 1))
    points = rotate_points(points, _axis_q([0.0, 0.0, -1.0], -elevs))
    return points * dists.view(-1, 1, 1)


# --------------------------------------------------------------------------
# Chamfer                                      (modules/loss/chamfer_distance.py)
# --------------------------------------------------------------------------
def chamfer_nn(p1, p2):
    """chamfer_distance.py:14-23 keeping the argmin the reference discards.
    Returns (min1 (B,N), idx1 (B,N), min2 (B,M), idx2 (B,M)); ties -> lowest
    index (torch.min(dim) semantics), distances are NON-squared."""
    diff = p1[:, :, None, :] - p2[:, None, :, :]       # :14
    dist = torch.sum(diff * diff, dim=3)               # :15
    dist1 = torch.sqrt(dist)                           # :19
    dist2 = torch.sqrt(torch.transpose(dist, 1, 2))    # :17,:20
    m1, i1 = torch.min(dist1, dim=2)                   # :22
    m2, i2 = torch.min(dist2, dim=2)                   # :23
    return m1, i1, m2, i2


def chamfer_nn_ieee(p1, p2):
    """Same as chamfer_nn but with an IEEE-754 correctly rounded sqrt (numpy).  torch.sqrt on
    CPU goes through MKL VML (HA mode, <= 1 ulp, NOT correctly rounded) for large contiguous
    tensors: about 0.6 % of its results are 1 ulp off and which ones depends on tensor size and
    CPU, so the reference's CPU distances are only defined to 1 ulp.  d2 (sub, mul, add) is exact
    in both; this variant is what the kernels are compared with bit-for-bit (it is also what the
    reference computes on its own CUDA device, where sqrt is correctly rounded)."""
    import numpy as np
    diff = p1[:, :, None, :] - p2[:, None, :, :]
    dist = torch.sum(diff * diff, dim=3)
    s = torch.from_numpy(np.sqrt(dist.numpy()))
    m1, i1 = torch.min(s, dim=2)
    m2, i2 = torch.min(torch.transpose(s, 1, 2), dim=2)
    return m1, i1, m2, i2


def chamfer_loss(p1, p2, each_batch=False, w1=1.0, w2=1.0):
    """ChamferDistanceLoss.forward (chamfer_distance.py:10-30), same expression
    (dense B*N*M, differentiable through torch autograd exactly like the
    reference)."""
    m1, _, m2, _ = chamfer_nn(p1, p2)
    loss = w1 * m1.mean(1) + w2 * m2.mean(1)           # :25-28
    return loss if each_batch else loss.mean()         # :30


def chamfer_loss_chunked(p1, p2, w1=1.0, w2=1.0, chunk=4):
    """Same loss, batch processed in chunks so the dense (B,N,M,3) tensor of
    chamfer_distance.py:14 fits in memory (12.9 GB at config 3).  Returns the
    per-sample loss (B,)."""
    return torch.cat([chamfer_loss(p1[i:i + chunk], p2[i:i + chunk], True, w1, w2)
                      for i in range(0, p1.shape[0], chunk)])


def vp_diverse_loss(translates, gt_points):
    """VPDiverseLoss.forward (vp_diverse.py:12-18): Chamfer between the K
    primitive centres and the GT points with w1=0.5, w2=1.0."""
    centres = torch.cat([t[:, None, :] for t in translates], 1)       # :15
    return chamfer_loss(centres, gt_points, w1=0.5, w2=1.0)           # :17


# --------------------------------------------------------------------------
# Philox4x32-10 (counter-based RNG used by the HIP sampler in production mode)
# --------------------------------------------------------------------------
def philox4x32_10(counter, key):
    """numpy restatement of Philox4x32-10 (Salmon et al., SC'11).  counter
    (...,4) uint32, key (...,2) uint32 -> (...,4) uint32."""
    import numpy as np
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0 = key[..., 0].astype(np.uint64)
    k1 = key[..., 1].astype(np.uint64)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return np.stack([x.astype(np.uint32) for x in c], -1)


def philox_uniforms(seed, sample_base, B, K, n):
    """Uniform draws the HIP sampler generates in production mode.  One Philox
    call per point: counter = (point index, primitive index, GLOBAL sample index
    = sample_base + b, 0), key = (seed lo, seed hi); u_i = (x_i >> 8) * 2^-24
    (24-bit mantissa, in [0,1)).  Keyed on the global sample index so results do
    not depend on how the batch is sharded over ranks.
    Returns float32 tensor (B,K,n,3)."""
    import numpy as np
    b = (np.arange(B, dtype=np.uint64) + np.uint64(sample_base))[:, None, None]
    k = np.arange(K, dtype=np.uint32)[None, :, None]
    p = np.arange(n, dtype=np.uint32)[None, None, :]
    ctr = np.zeros((B, K, n, 4), dtype=np.uint32)
    ctr[..., 0] = p
    ctr[..., 1] = k
    ctr[..., 2] = (b & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 3] = (b >> np.uint64(32)).astype(np.uint32)
    key = np.zeros((B, K, n, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    x = philox4x32_10(ctr, key)
    u = (x[..., :3] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return torch.from_numpy(u)


# --------------------------------------------------------------------------
# soft raster  (NEW OPERATOR - parity unpinned, see module docstring)
# --------------------------------------------------------------------------
FOVY_DEG = 49.13434207744484   # kaolin v0.1 DIBRenderer default field of view (recalled)
X_CLAMP = 80.0                 # clamp of the coverage logit
E_CLAMP = 8.0                  # clamp of the soft-min depth logit
MESH_MIN_AREA2 = 1e-12         # twice the NDC area below which a face of mesh_raster has no inside
EPS_H = 1e-3                   # squareplus smoothing of relu(1 - m2) inside the chord-length sqrt
DELTA_S0 = 1e-12               # guard of the soft-min normaliser
EPS_D = 1e-9                   # guard for ray components parallel to a box face


def camera_basis(cam, dtype=torch.float32):
    """Look-at camera of the reference's render call sites: eye at
    dist*(cos(elev)cos(azim), sin(elev), cos(elev)sin(azim)) looking at the
    origin, up +Y (vertex_renderer.py:18 set_look_at_parameters([azim],[elev],
    [dist]), angles in degrees; view-centred training uses dist=1, elev=azim=0:
    train.py:172-174).  For that view image-x is -z and image-y is -y, which is
    the convention modules/network/gcn.py:152-153 relies on.
    cam (B,3) = (dist, elev_deg, azim_deg).  Returns eye, right, up, fwd (B,3)."""
    cam = cam.to(dtype)
    d, el, az = cam[:, 0], cam[:, 1] * (math.pi / 180), cam[:, 2] * (math.pi / 180)
    eye = torch.stack([d * torch.cos(el) * torch.cos(az), d * torch.sin(el),
                       d * torch.cos(el) * torch.sin(az)], 1)
    zax = eye / eye.norm(dim=1, keepdim=True)            # camera +Z (points away from the scene)
    yup = torch.tensor([[0.0, 1.0, 0.0]], dtype=dtype).expand_as(zax)
    right = torch.cross(yup, zax, dim=1)
    right = right / right.norm(dim=1, keepdim=True)
    up = torch.cross(zax, right, dim=1)
    return eye, right, up, -zax


def pixel_grid(H, W, dtype=torch.float32):
    """Pixel-centre ray slopes: px (W,), py (H,).  Ray = fwd + px*right + py*up,
    so the ray parameter is z-depth along the optical axis."""
    th = math.tan(0.5 * FOVY_DEG * math.pi / 180)
    px = ((2 * (torch.arange(W, dtype=dtype) + 0.5) / W) - 1) * (th * W / H)
    py = (1 - (2 * (torch.arange(H, dtype=dtype) + 0.5) / H)) * th
    return px, py


def raster(params, types, cam, H, W, sigma=0.05, gamma=0.1, z_far=2.0):
    """Soft raster of K volumetric primitives per image.

    params (B,K,10) = (v3 semi-axes / half extents, q4 axis-angle, t3 centre),
    types list[K], cam (B,3).  Returns alpha (B,H,W) and depth (B,H,W).

    Per pixel ray o + s*d (o = eye, d = fwd + px*right + py*up) and primitive k,
    in the primitive's scaled frame  o~ = R^T(o - t)/v,  d~ = R^T d / v :
      ellipsoid:  s* = -(o~.d~)/(d~.d~),  w = o~ + s* d~,  m2 = w.w  (squared miss
                  distance of the line from the unit sphere's centre);
                  z = s* - sqrt(h/(d~.d~)),  h = squareplus(1-m2) = (u + sqrt(u^2+EPS_H))/2
                  (smooth relu: the entry depth inside, ~s* outside, no kink at the
                  silhouette edge - a relu there makes the gradient jump by 0.5/sqrt(eps)
                  exactly where fp32 cannot resolve the sign of 1-m2)
      cuboid:     lam = max over axis pairs (i<j) of |o~_j d~_i - o~_i d~_j| /
                  (|d~_i|+|d~_j|)  (smallest inflation of the unit box the line
                  touches; first pair wins ties), m2 = lam^2;
                  z = max_i ( -(L*sign(d~_i) + o~_i)/d~_i ),  L = max(lam,1)
      coverage    x = clamp((1-m2)/sigma, -X_CLAMP, X_CLAMP),  a = sigmoid(x)
      soft-min    e = clamp((z_ref - z)/gamma, -E_CLAMP, E_CLAMP), wgt = a*exp(e),
                  z_ref = camera distance
    composite:    alpha = 1 - prod_k (1-a_k)
                  zbar  = sum_k wgt_k z_k / (sum_k wgt_k + DELTA_S0)
                  depth = z_far + alpha * (zbar - z_far)
    """
    dt = params.dtype
    B, K, _ = params.shape
    eye, right, up, fwd = camera_basis(cam, dt)
    px, py = pixel_grid(H, W, dt)
    R = rotation_matrices(params[:, :, 3:7].reshape(B * K, 4)).reshape(B, K, 3, 3)
    v = params[:, :, 0:3]
    t = params[:, :, 7:10]
    Rt = R.transpose(2, 3)
    ot = torch.einsum('bkij,bkj->bki', Rt, eye[:, None, :] - t) / v                 # (B,K,3)
    Mr = torch.einsum('bkij,bj->bki', Rt, right) / v
    Mu = torch.einsum('bkij,bj->bki', Rt, up) / v
    Mf = torch.einsum('bkij,bj->bki', Rt, fwd) / v
    # d~ (B,K,H,W,3)
    dtl = (Mf[:, :, None, None, :] + px[None, None, None, :, None] * Mr[:, :, None, None, :]
           + py[None, None, :, None, None] * Mu[:, :, None, None, :])
    o = ot[:, :, None, None, :]
    is_box = torch.tensor([tp == CUBOID for tp in types])[None, :, None, None]

    # ellipsoid
    A = (dtl * dtl).sum(-1)
    Bq = (o * dtl).sum(-1)
    s_star = -Bq / A
    wv = o + s_star[..., None] * dtl
    m2_s = (wv * wv).sum(-1)
    uu = 1 - m2_s
    rr = torch.sqrt(uu * uu + EPS_H)
    # squareplus(u) = (u + sqrt(u^2 + eps)) / 2, written without cancellation for u < 0
    hh = torch.where(uu >= 0, 0.5 * (uu + rr), (0.5 * EPS_H) / (rr - uu))
    z_s = s_star - torch.sqrt(hh / A)

    if bool(is_box.any()):
        ad = dtl.abs()
        sgn = torch.where(dtl < 0, -torch.ones_like(dtl), torch.ones_like(dtl))
        dsafe = torch.where(ad < EPS_D, sgn * EPS_D, dtl)
        ox, oy, oz = o[..., 0], o[..., 1], o[..., 2]
        dx, dy, dz = dtl[..., 0], dtl[..., 1], dtl[..., 2]
        lam01 = (oy * dx - ox * dy).abs() / (ad[..., 0] + ad[..., 1] + EPS_D)
        lam02 = (oz * dx - ox * dz).abs() / (ad[..., 0] + ad[..., 2] + EPS_D)
        lam12 = (oz * dy - oy * dz).abs() / (ad[..., 1] + ad[..., 2] + EPS_D)
        lam = torch.where(lam02 > lam01, lam02, lam01)
        lam = torch.where(lam12 > lam, lam12, lam)
        L = torch.clamp(lam, min=1.0)
        tn = -(L[..., None] * sgn + o) / dsafe
        z_b = torch.where(tn[..., 1] > tn[..., 0], tn[..., 1], tn[..., 0])
        z_b = torch.where(tn[..., 2] > z_b, tn[..., 2], z_b)
        m2 = torch.where(is_box, lam * lam, m2_s)
        z = torch.where(is_box, z_b, z_s)
    else:
        m2, z = m2_s, z_s

    x = torch.clamp((1 - m2) / sigma, -X_CLAMP, X_CLAMP)
    ex = torch.exp(-x.abs())
    a = torch.where(x >= 0, 1 / (1 + ex), ex / (1 + ex))
    c = torch.where(x >= 0, ex / (1 + ex), 1 / (1 + ex))
    z_ref = cam[:, 0].to(dt)[:, None, None, None]
    e = torch.clamp((z_ref - z) / gamma, -E_CLAMP, E_CLAMP)
    wgt = a * torch.exp(e)
    alpha = 1 - torch.prod(c, dim=1)
    zbar = (wgt * z).sum(1) / (wgt.sum(1) + DELTA_S0)
    depth = z_far + alpha * (zbar - z_far)
    return alpha, depth


# ----------------------------------------------------------------------------------------------------------------
# Triangle-mesh path: a mesh that carries no primitives (train_sphere.py:53-76,128: 386.obj deformed in place,
# sampled with kaolin's TriangleMesh.sample and rasterised by kaolin's DIBRenderer through vertex_renderer.py:20-24).
# kaolin is absent, so neither function below has reference outputs: PARITY UNPINNED, the call-site contract
# (shapes, camera convention, "alpha is a soft silhouette of the union of the triangles", "samples are uniform on the
# surface") is what they restate; the HIP kernels are held to these functions.
MESH_NEAR = 1e-3          # faces with a vertex closer than this to the camera plane are not drawn


def mesh_project(verts, cam):
    """verts (B,P,3), cam (B,3) -> (B,P,3): x, y in NDC (y in [-1,1] over the image height, x in [-W/H, W/H]; +x to
    the right, +y up: the pixel grid of `pixel_grid` divided by tan(fov/2)) and z = depth along the optical axis."""
    dt = verts.dtype
    eye, right, up, fwd = camera_basis(cam, dt)
    rel = verts - eye[:, None, :]
    xc = (rel * right[:, None, :]).sum(-1)
    yc = (rel * up[:, None, :]).sum(-1)
    zc = (rel * fwd[:, None, :]).sum(-1)
    th = math.tan(0.5 * FOVY_DEG * math.pi / 180)
    zs = torch.where(zc > MESH_NEAR, zc, torch.ones_like(zc))          # not drawn anyway: keeps the division finite
    return torch.stack([xc / (zs * th), yc / (zs * th), zc], -1)


def mesh_raster(verts, faces, cam, H, W, sigma=1e-4):
    """Soft silhouette of a triangle mesh: alpha (B,H,W).
    Per pixel centre p (NDC) and face f with projected corners (a, b, c):
        d2_f = min over the three edges of the squared distance from p to the SEGMENT;
        s_f = +1 if p lies inside the triangle (either winding: no back-face culling, as a silhouette has none) else -1;
        a_f = sigmoid(s_f * d2_f / sigma);      alpha = 1 - prod_f (1 - a_f)
    (the probabilistic union of SoftRas / DIB-R's soft alpha: the 0.5 contour of a_f is the triangle's outline;
    sigma in NDC^2: 1e-4 is about one pixel of softness at 128 x 128).  Faces with a vertex at depth <= MESH_NEAR are
    skipped.  Differentiable w.r.t. verts; vectorised over pixels x faces, so small cases only."""
    dt = verts.dtype
    B = verts.shape[0]
    pr = mesh_project(verts, cam)                                        # (B,P,3)
    th = math.tan(0.5 * FOVY_DEG * math.pi / 180)
    px, py = pixel_grid(H, W, dt)
    gx = (px / th)[None, :].expand(H, W).reshape(-1)                      # (HW,)
    gy = (py / th)[:, None].expand(H, W).reshape(-1)
    tri = pr[:, faces.long(), :]                                         # (B,F,3,3)
    ok = (tri[..., 2] > MESH_NEAR).all(-1)                               # (B,F)
    ax, ay = tri[:, :, 0, 0], tri[:, :, 0, 1]
    bx, by = tri[:, :, 1, 0], tri[:, :, 1, 1]
    cx, cy = tri[:, :, 2, 0], tri[:, :, 2, 1]

    def seg_d2(x0, y0, x1, y1):                                          # -> (B,F,HW)
        ex, ey = (x1 - x0)[..., None], (y1 - y0)[..., None]
        wx, wy = gx[None, None, :] - x0[..., None], gy[None, None, :] - y0[..., None]
        t = ((wx * ex + wy * ey) / (ex * ex + ey * ey).clamp_min(1e-20)).clamp(0.0, 1.0)
        qx, qy = wx - t * ex, wy - t * ey
        return qx * qx + qy * qy

    def edge(x0, y0, x1, y1):                                            # 2D cross (b - a) x (p - a)
        return (x1 - x0)[..., None] * (gy[None, None, :] - y0[..., None]) - (y1 - y0)[..., None] * (gx[None, None, :] - x0[..., None])
    d2 = torch.minimum(torch.minimum(seg_d2(ax, ay, bx, by), seg_d2(bx, by, cx, cy)), seg_d2(cx, cy, ax, ay))
    e0, e1, e2 = edge(ax, ay, bx, by), edge(bx, by, cx, cy), edge(cx, cy, ax, ay)
    # e0 + e1 + e2 = twice the signed area: a face of (next to) no area has no inside (it used to be inside everywhere)
    inside = (((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))) & (((e0 + e1) + e2).abs() > MESH_MIN_AREA2)
    logit = torch.where(inside, d2, -d2) / sigma
    a = torch.sigmoid(logit.clamp(-80.0, 80.0))
    a = torch.where(ok[..., None], a, torch.zeros_like(a))
    alpha = 1.0 - torch.prod(1.0 - a, dim=1)
    return alpha.reshape(B, H, W)


def mesh_face_normals(verts, faces):
    """Unit face normals (B,F,3): the third output of vertex_renderer.py:24 (DIBRenderer returns face normals)."""
    tri = verts[:, faces.long(), :]
    n = torch.cross(tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0], dim=-1)
    return n / n.norm(dim=-1, keepdim=True).clamp_min(1e-20)


def mesh_sample(verts, faces, u):
    """Area-weighted uniform samples on a triangle mesh (what train_sphere.py:76 asks of kaolin's TriangleMesh.sample):
    verts (P,3), faces (F,3), u (n,3) uniforms in [0,1) -> points (n,3), face index (n,).
    Face from u[:,0] through the cumulative face areas (fp32 inclusive prefix sums in face order, the face is the first
    one whose cumulative area exceeds u0 * total); barycentric point from (u1, u2) with the square-root map
    p = (1 - r) a + r (1 - u2) b + r u2 c, r = sqrt(u1), which is uniform over the triangle.  The face choice is not
    differentiated (the areas enter through a discrete choice); the points are linear in the vertices."""
    f = faces.long()
    a, b, c = verts[f[:, 0]], verts[f[:, 1]], verts[f[:, 2]]
    area = 0.5 * torch.cross(b - a, c - a, dim=1).norm(dim=1).detach().float()
    cdf = torch.cumsum(area, 0)
    target = u[:, 0].float() * cdf[-1]
    idx = torch.searchsorted(cdf, target, right=True).clamp_max(f.shape[0] - 1)
    r = torch.sqrt(u[:, 1]).to(verts.dtype)
    w0, w1, w2 = 1.0 - r, r * (1.0 - u[:, 2].to(verts.dtype)), r * u[:, 2].to(verts.dtype)
    pts = w0[:, None] * a[idx] + w1[:, None] * b[idx] + w2[:, None] * c[idx]
    return pts, idx


def philox_uniforms_mesh(seed, mesh_index, n):
    """The draws of vpn_mesh_sample for one mesh: point i uses counter (i, 0xFFFFFFFF, mesh_index) of the same
    Philox4x32-10 stream as the primitive sampler (primitive slot 0xFFFFFFFF is never a primitive)."""
    import numpy as np
    i = np.arange(n, dtype=np.uint32)
    ctr = np.stack([i, np.full(n, 0xFFFFFFFF, np.uint32), np.full(n, mesh_index & 0xFFFFFFFF, np.uint32),
                    np.full(n, (mesh_index >> 32) & 0xFFFFFFFF, np.uint32)], 1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    out = philox4x32_10(ctr, key)
    return torch.from_numpy(((out[:, :3] >> 8).astype(np.float32) * np.float32(2.0 ** -24)))


def silhouette_loss(alpha, gt_silhouettes, func='L1'):
    """SilhouetteLoss.forward (modules/loss/silhouette.py:13-23): L1Loss / MSELoss
    (mean) between the predicted alpha (B,1,H,W) and the GT silhouette."""
    pred = alpha[:, None, :, :]
    d = pred - gt_silhouettes
    return d.abs().mean() if func == 'L1' else (d * d).mean()


# ----------------------------------------------------------------------------- EMD (auction), row f1
# PARITY UNPINNED against the reference's CUDA extension (modules/loss/emd/emd_cuda.cu): it needs nvcc, ships no
# stored outputs, and its own check (test_emd, emd_module.py:81-95) only re-derives the distance from the
# assignment.  What follows restates its rounds with the races resolved deterministically (lowest index wins);
# tests pin it to an exact assignment solver (scipy linear_sum_assignment) through the auction's eps-optimality.

def emd_auction(xyz1, xyz2, eps, iters):
    """emd_cuda_forward (emd_cuda.cu:228-282) for one batch: xyz1, xyz2 (B,n,3) fp32 ->
    dist (B,n) fp32 squared distance to the assigned point, assignment (B,n) int32.

    Per iteration (:246-273): list the unassigned points; each bids for the target maximising
    3 - |x1-x2| - price (Bid :95-179, value :143, increment best-second+eps :175-176, atomicMax :177);
    the bidder with the target's maximum increment holds it (GetMax :181-194);
    holders take the target, evict the previous owner, raise the price (Assign :196-215); on the last
    iteration every bidder is assigned to its bid.  fp32 throughout, sums as ((dx2+dy2)+dz2), no FMA."""
    import numpy as np
    x1 = np.ascontiguousarray(xyz1.detach().cpu().numpy() if hasattr(xyz1, 'detach') else xyz1, dtype=np.float32)
    x2 = np.ascontiguousarray(xyz2.detach().cpu().numpy() if hasattr(xyz2, 'detach') else xyz2, dtype=np.float32)
    B, n, _ = x1.shape
    assert x2.shape == x1.shape and iters >= 1
    eps = np.float32(eps)
    dist = np.zeros((B, n), np.float32)
    assignment = np.full((B, n), -1, np.int32)
    for b in range(B):
        a, c = x1[b], x2[b]
        assign = np.full(n, -1, np.int64)            # emd_module.py:44-50
        assign_inv = np.full(n, -1, np.int64)
        price = np.zeros(n, np.float32)
        for it in range(iters):
            last = it == iters - 1
            U = np.nonzero(assign == -1)[0]
            if U.size == 0:
                break
            best = np.empty(U.size, np.float32)
            second = np.empty(U.size, np.float32)
            bid = np.empty(U.size, np.int64)
            for s in range(0, U.size, 256):          # chunks bound the (bidders x targets) temporaries
                u = U[s:s + 256]
                dx = c[None, :, 0] - a[u, None, 0]
                dy = c[None, :, 1] - a[u, None, 1]
                dz = c[None, :, 2] - a[u, None, 2]
                val = (np.float32(3.0) - np.sqrt(((dx * dx) + (dy * dy)) + (dz * dz))) - price[None, :]
                bi = val.argmax(1)                   # first (lowest) index of the maximum, :144
                rows = np.arange(u.size)
                bv = val[rows, bi]
                val[rows, bi] = -np.inf
                sv = val.max(1) if n > 1 else np.full(u.size, -1e9, np.float32)
                best[s:s + 256], second[s:s + 256], bid[s:s + 256] = bv, np.maximum(sv, np.float32(-1e9)), bi
            inc = (best - second) + eps              # :175-176
            if last:
                assign[U] = bid                      # :203 with `last`
                break
            # :177 atomicMax + GetMax :181-194: the largest increment holds the target.  The reference lets every
            # bidder within 1e-6 of the maximum store its index and keeps whichever store lands last; the
            # exact maximum, lowest bidder among equal ones, is one of those outcomes and is what is fixed here.
            order = np.lexsort((U, -inc.astype(np.float64), bid))
            first = np.unique(bid[order], return_index=True)[1]
            win = order[first]
            wi, wt = U[win], bid[win]
            prev = assign_inv[wt]
            assign[prev[prev != -1]] = -1            # :206-207
            assign_inv[wt] = wi
            assign[wi] = wt
            price[wt] = price[wt] + inc[win]         # :211
        d = a - c[assign]
        dist[b] = ((d[:, 0] * d[:, 0]) + (d[:, 1] * d[:, 1])) + (d[:, 2] * d[:, 2])     # CalcDist :217-226
        assignment[b] = assign
    return torch.from_numpy(dist), torch.from_numpy(assignment)


def emd_backward(xyz1, xyz2, grad_dist, assignment):
    """NmDistanceGradKernel (emd_cuda.cu:284-300): grad_xyz1 = (2 g)(x1 - x2[assignment]); none for xyz2."""
    idx = assignment.long()[..., None].expand(-1, -1, 3)
    return (grad_dist * 2)[..., None] * (xyz1 - torch.gather(xyz2, 1, idx))


# ----------------------------------------------------------------------------- head post-processing, row f4
# PARITY UNPINNED in the strict sense: modules/network/vpnet_one_resnet.py imports torchvision (absent here), so its
# static methods cannot be run; the three lines they consist of are restated below (torch's own sigmoid / tanh /
# clamp, so the arithmetic is the reference's).

def head_post_process(volumes, rotates, translates, is_sigmoid=True, clamp_min=0.01, clamp_max=0.8,
                      volume_restrict=(8, 10, 10)):
    """restrict_range (vpnet_one_resnet.py:67-77) -> split(3|4|3, dim=1) (:36-38) -> restrict_volumes (:79-85),
    returned packed as (B,K,10) = (v, q, t) per primitive (the layout of sample_primitives / raster above)."""
    if is_sigmoid:
        v = torch.sigmoid(volumes) + 0.1
        q = torch.sigmoid(rotates)
        t = torch.tanh(translates)
    else:
        v = torch.clamp(volumes, min=clamp_min + 1e-8, max=clamp_max)
        q = torch.clamp(rotates, min=-1, max=1)
        t = torch.clamp(translates, min=-1, max=1)
    B = volumes.shape[0]
    K = volumes.shape[1] // 3
    v = v.reshape(B, K, 3) / torch.tensor(list(volume_restrict), dtype=v.dtype)
    return torch.cat([v, q.reshape(B, K, 4), t.reshape(B, K, 3)], dim=2)


def train_step(params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, kinds, n, H, W, weights, seed,
               eps=0.005, iters=50, sigma=0.05, gamma=0.1, z_far=2.0):
    """One training iteration's loss of the reference (train.py:243-262) and its gradient w.r.t. the packed primitive
    parameters, fp32 on the CPU: sampler (Philox replay of the kernel's draws) -> view-centred Chamfer (train.py:160) +
    object-centred Chamfer through view_to_obj_points (:158-161) + L1 silhouette loss with the view-centred camera
    (:169-176, silhouette.py:13-23) + VP-diversity (:185, vp_diverse.py:15-17) + sqrt(EMD dist).mean() (:193-195; the
    assignment is a constant of the graph, as in emdFunction, emd_module.py:58-70).  weights = (L_VIEW_CD, L_CAN_CD, L_SIL,
    L_VP_DIV, L_EMD).  Returns ([6] = the five weighted terms and their sum, d total / d params)."""
    B, K = params.shape[0], params.shape[1]
    w = [float(x) for x in weights]
    p = params.detach().clone().requires_grad_(True)
    u = philox_uniforms(seed, 0, B, K, n)
    pred = sample_primitives(p, kinds, u)
    zero = torch.zeros(())
    view_cd = chamfer_loss(pred, gt_view) * w[0]
    obj_cd = chamfer_loss(view_to_obj_points(pred, dists, elevs, azims, angles), gt_canon) * w[1]
    sil = zero
    if w[2]:
        cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(B, 3)                        # train.py:172-174
        alpha, _ = raster(p, kinds, cam, H, W, sigma, gamma, z_far)
        sil = (alpha - gt_sil.reshape(B, H, W)).abs().mean() * w[2]
    div = chamfer_loss(p[:, :, 7:10], gt_view, w1=0.5, w2=1.0) * w[3] if w[3] else zero
    emd = zero
    if w[4]:
        _, assign = emd_auction(pred.detach(), gt_view, eps, iters)
        picked = torch.gather(gt_view, 1, assign.long()[..., None].expand(-1, -1, 3))
        emd = torch.sqrt(((pred - picked) ** 2).sum(-1)).mean() * w[4]
    total = view_cd + obj_cd + sil + div + emd
    total.backward()
    return torch.stack([view_cd, obj_cd, sil, div, emd, total]).detach(), p.grad
