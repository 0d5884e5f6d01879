"""modules/loss of the reference (chamfer_distance.py, vp_diverse.py, silhouette.py, emd/emd_module.py)
on the HIP Chamfer, raster and auction kernels.  Same class names and forward signatures."""
import torch
import torch.nn as nn

from .. import config
from ..ops import ChamferFunction, EmdFunction, RasterTotalFunction
from ..primitives import PrimitivePack
from .render import VertexRenderer


class ChamferDistanceLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, points1: torch.Tensor, points2: torch.Tensor, each_batch=False,
                w1=config.CD_W1, w2=config.CD_W2) -> torch.Tensor:
        """chamfer_distance.py:10-30: w1*mean_i min_j|a_i-b_j| + w2*mean_j min_i|a_i-b_j|
        (non-squared), per sample if each_batch else the batch mean."""
        self.check_parameters(points1)
        self.check_parameters(points2)
        loss = ChamferFunction.apply(points1, points2, w1, w2)
        return loss if each_batch else loss.mean()

    @staticmethod
    def check_parameters(points: torch.Tensor):
        assert points.ndimension() == 3  # (B, N, 3)        chamfer_distance.py:32-35
        assert points.size(-1) == 3


class EarthMoverDistanceLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, input1: torch.Tensor, input2: torch.Tensor, eps, iters):
        """emd_module.py:77-78: (dist [B,n], assignment [B,n] int32); train.py:193 calls it with
        eps=0.005, iters=50 on point sets normalised to the unit cube."""
        return EmdFunction.apply(input1, input2, eps, iters)


class VPDiverseLoss(nn.Module):
    def __init__(self, vp_num=None):
        super().__init__()
        self.cd_loss_func = ChamferDistanceLoss()
        self.vp_num = vp_num             # the reference asserts len == config.VP_NUM (vp_diverse.py:23)

    def forward(self, translates: list, gt_points: torch.Tensor) -> torch.Tensor:
        """vp_diverse.py:12-18."""
        self.check_parameters(translates)
        vp_center_points = torch.cat([t[:, None, :] for t in translates], 1)
        return self.cd_loss_func(vp_center_points, gt_points, w1=0.5, w2=1.0)

    def check_parameters(self, translates):
        assert isinstance(translates, list)
        if self.vp_num is not None:
            assert len(translates) == self.vp_num


class SilhouetteLoss(nn.Module):
    def __init__(self, loss_func=config.SILHOUETTE_LOSS_FUNC):
        super().__init__()
        self.loss_func = nn.L1Loss() if loss_func == 'L1' else nn.MSELoss()     # silhouette.py:11
        self.is_mse = loss_func != 'L1'

    def forward(self, predict_meshes, gt_silhouettes: torch.Tensor,
                dists: torch.Tensor, elevs: torch.Tensor, azims: torch.Tensor) -> torch.Tensor:
        """silhouette.py:13-23.  `predict_meshes` is the reference's list of B composed meshes (train.py:146-149,
        made by Meshing: each carries its primitives), a list of per-sample packs, or one PrimitivePack for the
        batch; the B sequential
        renders of silhouette.py:16-18 become one launch at the GT silhouette's resolution."""
        H, W = gt_silhouettes.shape[-2:]
        try:
            predict_meshes = PrimitivePack.of(predict_meshes)
        except TypeError:
            # meshes without (valid) primitives, e.g. train_sphere.py:119-128 (a sphere mesh deformed in place): their
            # triangles are rendered, silhouette.py:16-22 as written (B renders -> cat -> L1Loss / MSELoss)
            alpha, _ = VertexRenderer.triangle_alpha(predict_meshes, dists, elevs, azims, H, W)
            return self.loss_func(alpha[:, None], gt_silhouettes.to(alpha.device).float().reshape(alpha.shape[0], 1, H, W))
        B = len(predict_meshes)
        dev = predict_meshes.params.device
        cam = torch.stack([dists.to(dev).float().reshape(-1).expand(B), elevs.to(dev).float().reshape(-1).expand(B),
                           azims.to(dev).float().reshape(-1).expand(B)], 1)
        # render + loss + gradient partials in one pass (silhouette.py:16-22 renders B times, concatenates and applies
        # L1Loss/MSELoss); with w_sil = 1, w_dep = 0 the differentiable total IS the silhouette loss
        losses = RasterTotalFunction.apply(predict_meshes.params, predict_meshes.kinds, cam, gt_silhouettes, None, H, W,
                                           VertexRenderer.sigma, VertexRenderer.gamma, VertexRenderer.z_far, self.is_mse,
                                           1.0, 0.0)
        return losses[2]
