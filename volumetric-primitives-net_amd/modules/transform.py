"""modules/transform of the reference (transform.py, rotate.py, translate.py) on the HIP
transform kernel.  Same function names, argument order and shape assertions."""
import torch

from ..ops import CameraTransformFunction, TransformFunction


def _check_points(points):
    assert points.ndimension() == 3      # (B, N, 3)   rotate.py:49-51
    assert points.size(-1) == 3


def rotate_points(points: torch.Tensor, quaternions: torch.Tensor):
    """rotate.py:7-25.  quaternions (B,4): [:, :3] axis (not normalised), [:, 3] angle in turns."""
    _check_points(points)
    assert quaternions.ndimension() == 2 and quaternions.size(-1) == 4      # rotate.py:54-56
    return TransformFunction.apply(points, quaternions, None)


def translate_points(points: torch.Tensor, translations: torch.Tensor):
    """translate.py:4-8 (a broadcast add: left to ATen)."""
    _check_points(points)
    assert translations.ndimension() == 2 and translations.size(-1) == 3
    return points + translations.unsqueeze(1)


def transform_points(points: torch.Tensor, q: torch.Tensor, t: torch.Tensor):
    """transform.py:6-9: rotate then translate, one fused launch."""
    _check_points(points)
    B = points.size(0)
    assert q.size() == (B, 4)            # transform.py:17-18
    assert t.size() == (B, 3)
    return TransformFunction.apply(points, q, t)


def _axis_q(axis, angles):
    a = torch.tensor([axis], dtype=torch.float32, device=angles.device).repeat(angles.size(0), 1)
    return torch.cat([a, angles.view(-1, 1)], 1)


def rotate_points_forward_x_axis(points: torch.Tensor, angles: torch.Tensor):
    """transform.py:76-94 (angles in degrees)."""
    assert points.ndimension() == 3
    assert angles.ndimension() == 1
    return rotate_points(points, _axis_q([1.0, 0.0, 0.0], angles.view(-1) / 360))


def _camera_is_data(*tensors):
    for t in tensors:
        if t.requires_grad:
            raise RuntimeError('camera arguments are dataset values here (dataset.py:145-165); gradients with respect '
                               'to them are not implemented')


def obj_to_view_points(points, dists, elevs, azims):
    """transform.py:50-73: two rotations and the division by dist, one fused launch."""
    assert points.ndimension() == 3
    assert dists.ndimension() == elevs.ndimension() == azims.ndimension() == 1
    _camera_is_data(dists, elevs, azims)
    return CameraTransformFunction.apply(points, dists, elevs, azims, None, False)


def view_to_obj_points(points, dists, elevs, azims, angles):
    """transform.py:21-47: three rotations and the scale by dist, one fused launch."""
    assert points.ndimension() == 3
    assert dists.ndimension() == elevs.ndimension() == azims.ndimension() == 1
    _camera_is_data(dists, elevs, azims, angles)
    return CameraTransformFunction.apply(points, dists, elevs, azims, angles, True)
