"""A training step of the reference (train.py:221-268) on the drop-in surface, with a stand-in network.

    python examples/train_step.py [--steps 20]

The reference's ResNet-18 trunk is out of scope: a two-layer MLP on a random feature vector
produces the three head outputs (volumes [B,3K], rotates [B,4K], translates [B,3K]).  Everything after that is the
code path of the reference, through vpn_amd's mirror of its modules:

    head post-processing     vpnet_one_resnet.py:34-41   pack_head_outputs
    predicted points         train.py:105-120            Sampling.sample_primitives
    view-centred Chamfer     train.py:160                ChamferDistanceLoss
    object-centred Chamfer   train.py:158-161            view_to_obj_points + ChamferDistanceLoss
    silhouette loss          train.py:176                SilhouetteLoss
    VP diversity loss        train.py:185                VPDiverseLoss
    EMD loss                 train.py:193                EarthMoverDistanceLoss
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd  # noqa: E402


class Heads(nn.Module):
    """Stand-in for VPNetOneRes: features -> (volumes, rotates, translates) raw head outputs."""

    def __init__(self, feat, K):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(feat, 256), nn.ReLU())
        self.volume_fc, self.rotate_fc, self.translate_fc = nn.Linear(256, 3 * K), nn.Linear(256, 4 * K), nn.Linear(256, 3 * K)

    def forward(self, x):
        h = self.trunk(x)
        return self.volume_fc(h), self.rotate_fc(h), self.translate_fc(h)


def training_losses(net, feats, gt_points, gt_sil, dists, elevs, azims, angles, kinds, sample_num, weights, seed):
    """total loss of train.py:243-262 (w = (L_VIEW_CD, L_CAN_CD, L_SIL, L_VP_DIV, L_EMD)) and its parts."""
    K = len(kinds)
    params = vpn_amd.pack_head_outputs(*net(feats))                               # [B,K,10]
    volumes, rotates, translates = vpn_amd.split_primitives(params)
    pred = vpn_amd.Sampling.sample_primitives(params, kinds, sample_num, seed=seed)   # [B, K*n, 3], view-centred
    cd = vpn_amd.ChamferDistanceLoss()
    view_cd = cd(pred, gt_points)
    obj_cd = cd(vpn_amd.view_to_obj_points(pred, dists, elevs, azims, angles),
                vpn_amd.view_to_obj_points(gt_points, dists, elevs, azims, angles))
    sil = vpn_amd.SilhouetteLoss()(vpn_amd.PrimitivePack(params, kinds), gt_sil, dists, elevs, azims)
    div = vpn_amd.VPDiverseLoss(vp_num=K)(translates, gt_points)
    dist, _ = vpn_amd.EarthMoverDistanceLoss()(pred, gt_points, 0.005, 50)         # needs K*n == M (emd_module.py:36)
    emd = torch.sqrt(dist).mean()
    w = weights
    total = w[0] * view_cd + w[1] * obj_cd + w[2] * sil + w[3] * div + w[4] * emd
    return total, {'view_cd': view_cd, 'obj_cd': obj_cd, 'sil': sil, 'vp_div': div, 'emd': emd}


def training_losses_fused(net, feats, gt_points, gt_sil, dists, elevs, azims, angles, kinds, sample_num, weights, seed):
    """The same loss as ONE autograd node (vpn_amd.TrainStepLossFunction: 9 launches forward with the auction on a second
    stream, 1 backward).  Needs K * sample_num >= 512 sampled points against as many GT points."""
    params = vpn_amd.pack_head_outputs(*net(feats))                               # [B,K,10]
    size = gt_sil.shape[-1]
    gt_canon = vpn_amd.view_to_obj_points(gt_points, dists, elevs, azims, angles)
    view_cd, obj_cd, sil, div, emd, total = vpn_amd.TrainStepLossFunction.apply(
        params, kinds, gt_points, gt_canon, gt_sil, dists, elevs, azims, angles, sample_num, seed, 0, size, size, weights)
    return total, {'view_cd': view_cd, 'obj_cd': obj_cd, 'sil': sil, 'vp_div': div, 'emd': emd}   # the WEIGHTED terms


def make_batch(B, K, sample_num, size, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    M = K * sample_num
    feats = torch.randn(B, 64, generator=g).to(dev)
    # a target made of K random ellipsoids: its surface points and its silhouette
    v = (torch.rand(B, K, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    target = torch.cat([v, torch.rand(B, K, 4, generator=g), 0.3 * (torch.rand(B, K, 3, generator=g) * 2 - 1)], 2).to(dev)
    kinds = [vpn_amd.SPHERE] * K
    with torch.no_grad():
        gt_points = vpn_amd.Sampling.sample_primitives(target, kinds, sample_num, seed=99)
        dists = torch.ones(B, device=dev)
        elevs = torch.zeros(B, device=dev)
        azims = torch.zeros(B, device=dev)
        _, alpha, _ = vpn_amd.VertexRenderer.render(vpn_amd.PrimitivePack(target, kinds), dists, elevs, azims,
                                                    image_size=(size, size))
        gt_sil = (alpha.reshape(B, 1, size, size) > 0.5).float()
    return feats, gt_points, gt_sil, dists, elevs, azims, torch.zeros(B, device=dev), kinds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8)          # config.py:9
    ap.add_argument('--prims', type=int, default=16)         # config.py:34
    ap.add_argument('--sample-num', type=int, default=128)   # config.py:8
    ap.add_argument('--size', type=int, default=128)         # config.py:49
    ap.add_argument('--fused', action='store_true', help='the whole loss as one autograd node (TrainStepLossFunction)')
    args = ap.parse_args()
    losses = training_losses_fused if args.fused else training_losses
    dev = torch.device('cuda')
    torch.manual_seed(1234)
    batch = make_batch(args.batch, args.prims, args.sample_num, args.size, dev)
    net = Heads(64, args.prims).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    weights = (1.0, 1.0, 1.0, 0.1, 1.0)
    for it in range(args.steps):
        opt.zero_grad()
        total, parts = losses(net, *batch, args.sample_num, weights, seed=1000 + it)
        total.backward()
        opt.step()
        print('step %3d  total %.5f  ' % (it, float(total.detach())) + '  '.join('%s %.5f' % (k, float(v.detach())) for k, v in parts.items()),
              flush=True)


if __name__ == '__main__':
    main()
