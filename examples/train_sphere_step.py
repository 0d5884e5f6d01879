"""A training step of the reference's train_sphere.py (:104-134) on the drop-in surface, with a stand-in network.

    python examples/train_sphere_step.py [--steps 20] [--obj path/to/386.obj]

train_sphere.py deforms a template sphere mesh instead of predicting primitives: SDNet (a ResNet-18, out of scope:
DESIGN.md 7) outputs one offset per vertex, the 386-vertex sphere of `386.obj` is deformed IN PLACE, sampled, compared
with the ground-truth cloud and rendered against the ground-truth silhouette.  The lines below are the reference's,
on vpn_amd's mirror of kaolin's TriangleMesh and of its loss modules:

    load_sphere_meshes    train_sphere.py:50-59    TriangleMesh.from_obj(...).cuda()      (a procedural sphere without --obj)
    deform_meshes         train_sphere.py:62-68    meshes[b].vertices += vertices_offset[b]
    sample_points         train_sphere.py:71-81    meshes[b].sample(n)[0]                  vpn_mesh_sample_fwd/bwd
    Chamfer loss          train_sphere.py:121      ChamferDistanceLoss                     vpn_chamfer_fwd_ws / bwd
    silhouette loss       train_sphere.py:125-128  SilhouetteLoss(list of meshes, ...)     vpn_mesh_raster_fwd/bwd
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd  # noqa: E402
from vpn_amd.modules.meshing import uv_sphere  # noqa: E402


class Offsets(nn.Module):
    """Stand-in for SDNet (sdnet.py:17-22): features -> vertex offsets [B,P,3]."""

    def __init__(self, feat, P):
        super().__init__()
        self.P = P
        self.fc = nn.Sequential(nn.Linear(feat, 256), nn.ReLU(), nn.Linear(256, 3 * P))

    def forward(self, x):
        return 0.1 * torch.tanh(self.fc(x)).reshape(-1, self.P, 3)


def load_sphere_meshes(B, dev, obj=None, radius=0.25):
    """train_sphere.py:50-59: one fresh template per sample (the deformation is in place)."""
    meshes = []
    for _ in range(B):
        if obj:
            m = vpn_amd.TriangleMesh.from_obj(obj)
        else:
            v, f = uv_sphere(12, 24)                                   # 288 vertices, a closed surface
            m = vpn_amd.TriangleMesh(v * radius, f)
        meshes.append(m.to(dev))
    return meshes


def training_losses(net, feats, gt_points, gt_sil, dists, elevs, azims, sample_num, l_sil, obj=None):
    B = feats.shape[0]
    sphere_meshes = load_sphere_meshes(B, feats.device, obj)
    vertices_offset = net(feats)                                       # train_sphere.py:111
    for b in range(B):
        sphere_meshes[b].vertices += vertices_offset[b]                # :66, in place
    predict_points = torch.cat([sphere_meshes[b].sample(sample_num)[0][None] for b in range(B)], dim=0)   # :75-79
    cd_loss = vpn_amd.ChamferDistanceLoss()(predict_points, gt_points)                                    # :121
    sil_loss = vpn_amd.SilhouetteLoss()(sphere_meshes, gt_sil, dists, elevs, azims) * l_sil               # :128
    return cd_loss + sil_loss, {'cd': cd_loss, 'sil': sil_loss}


def make_batch(B, M, size, dev, seed=0):
    """A ground truth the template can reach: an ellipsoid's surface samples and its silhouette from the view-centred
    camera of train_sphere.py:125-127."""
    g = torch.Generator().manual_seed(seed)
    axes = 0.15 + 0.2 * torch.rand(B, 1, 3, generator=g)
    d = torch.randn(B, M, 3, generator=g)
    gt_points = (d / d.norm(dim=2, keepdim=True) * axes).to(dev)
    dists, elevs, azims = torch.ones(B, device=dev), torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    v, f = uv_sphere(12, 24)
    gt_meshes = [vpn_amd.TriangleMesh((v * axes[b]).to(dev), f.to(dev)) for b in range(B)]
    with torch.no_grad():
        alpha = vpn_amd.VertexRenderer.render(gt_meshes, dists, elevs, azims, image_size=(size, size))[1]
    gt_sil = (alpha[..., 0] > 0.5).float()[:, None]
    feats = torch.randn(B, 64, generator=g).to(dev)
    return feats, gt_points, gt_sil, dists, elevs, azims


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--obj', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda')
    torch.manual_seed(0)
    batch = make_batch(4, 2048, 64, dev)                              # BASELINE config C1: batch 4, 64 x 64
    P = vpn_amd.load_obj(args.obj)[0].shape[0] if args.obj else 288
    net = Offsets(64, P).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    for it in range(args.steps):
        opt.zero_grad()
        total, parts = training_losses(net, *batch, 1024, 1.0, args.obj)
        total.backward()
        opt.step()
        print('step %3d  total %.5f  cd %.5f  sil %.5f' % (it, float(total.detach()), float(parts['cd'].detach()), float(parts['sil'].detach())))
