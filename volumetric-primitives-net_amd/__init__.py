"""vpn_amd — MI355X-native volumetric-primitive hot path (sampler, Chamfer, soft raster).

Import as `vpn_amd` (repo-root alias module); the directory name is fixed by the project
layout.  Everything here runs on hand-written HIP kernels in libvpn_hip.so through the C
ABI of include/vpn_hip.h; there is no CPU fallback."""
from . import config
from .ops import (SPHERE, CUBOID, SampleFunction, TransformFunction, ChamferFunction, EmdFunction, HeadPackFunction, MeshFunction,
                  CameraTransformFunction, RasterFunction,
                  RasterLossFunction, RasterTotalFunction, HotPathLossFunction, TrainStepLossFunction, chamfer_nn, kinds_tensor)
from .primitives import PrimitivePack, pack_primitives, kinds_from_counts
from .modules import (Sampling, ChamferDistanceLoss, EarthMoverDistanceLoss, SilhouetteLoss, VPDiverseLoss, VertexRenderer,
                      transform_points, rotate_points, translate_points, view_to_obj_points,
                      obj_to_view_points, rotate_points_forward_x_axis, pack_head_outputs, split_primitives, Meshing, TriangleMesh, load_obj)
from . import modules
