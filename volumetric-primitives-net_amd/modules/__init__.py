"""Host-side mirror of the reference's `modules` package for the hot path only:
same import names as train.py:11-15 (`from modules.sampling import Sampling`, ...)."""
from .sampling import Sampling
from .loss import ChamferDistanceLoss, EarthMoverDistanceLoss, SilhouetteLoss, VPDiverseLoss
from .render import VertexRenderer
from .transform import (transform_points, rotate_points, translate_points, view_to_obj_points,
                        obj_to_view_points, rotate_points_forward_x_axis)
from .network import pack_head_outputs, split_primitives
from .meshing import Meshing, TriangleMesh, load_obj
from .dataset import parse_split_csv, parse_rendering_metadata, split_rgba
