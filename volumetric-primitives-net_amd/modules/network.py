"""The one piece of modules/network of the reference that sits on the hot path: the post-processing of the three
head outputs (vpnet_one_resnet.py:34-41, :67-85; identical in vpnet_two_resnet.py and sdnet.py).  The networks
themselves (ResNet-18 trunk, MLP heads) are out of scope."""
import torch

from .. import config
from ..ops import HeadPackFunction


def pack_head_outputs(volumes: torch.Tensor, rotates: torch.Tensor, translates: torch.Tensor,
                      is_sigmoid=config.IS_SIGMOID, clamp_min=config.VP_CLAMP_MIN, clamp_max=config.VP_CLAMP_MAX,
                      volume_restrict=config.VOLUME_RESTRICT) -> torch.Tensor:
    """restrict_range (:67-77) -> split (:36-38) -> restrict_volumes (:79-85) in one launch: raw head outputs
    volumes (B,3K), rotates (B,4K), translates (B,3K) -> packed parameters (B,K,10), differentiable."""
    return HeadPackFunction.apply(volumes, rotates, translates, is_sigmoid, clamp_min, clamp_max, volume_restrict)


def split_primitives(params: torch.Tensor):
    """Packed (B,K,10) -> the three python lists of K tensors (B,3), (B,4), (B,3) that the reference's model
    returns (:36-41) and train.py:105-120, :185 consume; views, no copy."""
    K = params.shape[1]
    return ([params[:, k, 0:3] for k in range(K)], [params[:, k, 3:7] for k in range(K)],
            [params[:, k, 7:10] for k in range(K)])
