"""World-space ray / primitive intersection used to pin the raster's geometry (tests only).

Deliberately NOT the raster's formulation (which works in the primitive's scaled frame with a miss distance m2):
here a pixel's ray is intersected with the primitive in world space -- the quadric (x-t)^T A^-T A^-1 (x-t) = 1,
A = R diag(v), for an ellipsoid; the slab test in the box's own frame for a cuboid -- in float64.  The only
pieces shared with the oracle are the reference-pinned pose (`rotation_matrices`, golden g6) and the look-at
camera, which `test_camera_matches_reference_view_transform` ties to the reference's obj_to_view_points."""
import math

import torch

from oracle import vpn_oracle as O


def project(points, cam):
    """points (B,N,3), cam (B,3) -> slopes x, y (B,N) and z-depth along the optical axis (B,N)."""
    eye, right, up, fwd = O.camera_basis(cam, points.dtype)
    rel = points - eye[:, None]
    z = (rel * fwd[:, None]).sum(-1)
    return (rel * right[:, None]).sum(-1) / z, (rel * up[:, None]).sum(-1) / z, z


def slopes_to_pixels(x, y, H, W):
    """Continuous pixel coordinates (col, row) of ray slopes; pixel centres sit at integer + 0.5."""
    th = math.tan(0.5 * O.FOVY_DEG * math.pi / 180)
    col = (x / (th * W / H) + 1) * 0.5 * W
    row = (1 - y / th) * 0.5 * H
    return col, row


def pixel_rays(cam, rows, cols, H, W):
    """World-space rays through pixel centres: eye (B,3), dir (B,N,3) with unit component along the optical axis."""
    eye, right, up, fwd = O.camera_basis(cam, torch.float64)
    th = math.tan(0.5 * O.FOVY_DEG * math.pi / 180)
    px = ((2 * (cols.double() + 0.5) / W) - 1) * (th * W / H)
    py = (1 - (2 * (rows.double() + 0.5) / H)) * th
    d = fwd[:, None] + px[..., None] * right[:, None] + py[..., None] * up[:, None]
    return eye, d


def hit_ellipsoid(eye, d, v, q, t):
    """Ray eye + s d against the ellipsoid with semi-axes v, pose (q, t).  Returns (discriminant / a^2 -- positive
    = hit, the relative margin of the decision -- and the entry parameter s, valid where hit)."""
    R = O.rotation_matrices(q.double())                       # (B,3,3), pinned by golden g6
    o = torch.einsum('bji,bj->bi', R, eye - t.double()) / v.double()
    dd = torch.einsum('bji,bnj->bni', R, d) / v.double()[:, None]
    a = (dd * dd).sum(-1)
    b = (dd * o[:, None]).sum(-1)
    c = (o * o).sum(-1)[:, None] - 1
    disc = b * b - a * c
    s = (-b - torch.sqrt(disc.clamp_min(0))) / a
    return disc / (a * a), s


def hit_box(eye, d, v, q, t):
    """Slab test against the box with half extents v, pose (q, t): (margin = s_exit - s_entry, entry s)."""
    R = O.rotation_matrices(q.double())
    o = torch.einsum('bji,bj->bi', R, eye - t.double())
    dd = torch.einsum('bji,bnj->bni', R, d)
    dd = torch.where(dd.abs() < 1e-300, torch.full_like(dd, 1e-300), dd)
    s1 = (-v.double()[:, None] - o[:, None]) / dd
    s2 = (v.double()[:, None] - o[:, None]) / dd
    lo = torch.minimum(s1, s2).max(-1)[0]
    hi = torch.maximum(s1, s2).min(-1)[0]
    return hi - lo, lo


def uv_sphere_386(radius=1.0):
    """The topology of the reference's 386.obj (train_sphere.py:53): a UV sphere of 16 rings x 24 segments plus the two
    poles = 386 vertices, 2 x 24 cap triangles + 15 x 24 x 2 = 768 faces; a closed surface."""
    import math
    rings, seg = 16, 24
    vs = [[0.0, radius, 0.0]]
    for r in range(rings):
        th = math.pi * (r + 1) / (rings + 1)
        for s_ in range(seg):
            ph = 2.0 * math.pi * s_ / seg
            vs.append([radius * math.sin(th) * math.cos(ph), radius * math.cos(th), radius * math.sin(th) * math.sin(ph)])
    vs.append([0.0, -radius, 0.0])
    south = len(vs) - 1
    fs = []
    for s_ in range(seg):
        fs.append([0, 1 + (s_ + 1) % seg, 1 + s_])
        base = 1 + (rings - 1) * seg
        fs.append([south, base + s_, base + (s_ + 1) % seg])
    for r in range(rings - 1):
        for s_ in range(seg):
            a, b = 1 + r * seg + s_, 1 + r * seg + (s_ + 1) % seg
            c, d = a + seg, b + seg
            fs += [[a, b, c], [b, d, c]]
    v, f = torch.tensor(vs, dtype=torch.float32), torch.tensor(fs, dtype=torch.int64)
    assert v.shape == (386, 3) and f.shape == (768, 3)
    return v, f
