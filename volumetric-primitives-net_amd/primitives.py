"""Packed primitive tensor shared by the sampler and the raster.

The reference carries K python lists of (B,3) volumes, (B,4) axis-angle rotations and
(B,3) translations (vpnet_one_resnet.py:28-43, train.py:117); the kernels want one
contiguous [B,K,10] tensor so a whole batch x primitive set is one coalesced read."""
import torch

from .ops import SPHERE, CUBOID, PARAM_STRIDE, kinds_tensor, kinds_host, faces_fingerprint


def pack_primitives(volumes, rotates, translates):
    """K lists of (B,3),(B,4),(B,3) (the network's output format) -> [B,K,10].  Differentiable."""
    assert len(volumes) == len(rotates) == len(translates)
    return torch.stack([torch.cat([v, q, t], 1) for v, q, t in zip(volumes, rotates, translates)], 1)


def kinds_from_counts(cuboid_num, sphere_num, cone_num=0):
    """Primitive order of train.py:106-116: cuboids first, then spheres (cones must be 0:
    sampling.py:39-45 and meshing.py:22-25 are stubs in the reference)."""
    if cone_num:
        raise ValueError('cone primitives are not implemented (reference stubs: sampling.py:39-45)')
    return [CUBOID] * cuboid_num + [SPHERE] * sphere_num


class PrimitivePack:
    """What the renderer consumes in place of the reference's kaolin TriangleMesh
    (vertex_renderer.py:16): params [B,K,10] (or [K,10] for one image) + kinds [K]."""

    def __init__(self, params, kinds):
        if params.dim() == 2:
            params = params[None]
        assert params.dim() == 3 and params.size(-1) == PARAM_STRIDE
        self.params = params
        self.kinds = kinds_tensor(kinds, params.device)
        assert self.kinds.numel() == params.size(1)

    @classmethod
    def from_lists(cls, volumes, rotates, translates, kinds):
        return cls(pack_primitives(volumes, rotates, translates), kinds)

    def __len__(self):
        return self.params.size(0)

    def __getitem__(self, b):
        """Per-sample view, so `predict_meshes[i]` of silhouette.py:17 keeps working."""
        return PrimitivePack(self.params[b], self.kinds)

    @staticmethod
    def stack(packs):
        # compared on the host (kinds_host: no device synchronisation; the B meshes of train.py:122-149 share one tensor)
        k0 = kinds_host(packs[0].kinds)
        for p in packs[1:]:
            if p.kinds is not packs[0].kinds and kinds_host(p.kinds) != k0:
                raise ValueError('all samples of a batch must hold the same primitive kinds in the same order')
        return PrimitivePack(torch.cat([p.params for p in packs], 0), packs[0].kinds)

    @staticmethod
    def of(obj):
        """PrimitivePack behind `obj`: a pack itself, a mesh produced by Meshing.*_meshing / compose_meshes (its
        `.primitives`), or a list of either, one entry per sample (the list train.py:122-149 builds and train.py:176
        hands to SilhouetteLoss).  Anything else - e.g. a triangle mesh without primitives - raises TypeError."""
        if isinstance(obj, PrimitivePack):
            return obj
        if isinstance(obj, (list, tuple)):
            if not obj:
                raise TypeError('empty list of meshes')
            return PrimitivePack.stack([PrimitivePack.of(o) for o in obj])
        prims = getattr(obj, 'primitives', None)      # None once the mesh's vertices were edited (TriangleMesh.primitives)
        if isinstance(prims, PrimitivePack):
            return prims
        raise TypeError('the primitive raster renders primitive parameters: pass a PrimitivePack or a mesh made by '
                        'Meshing.sphere_meshing / cuboid_meshing / compose_meshes whose vertices are unchanged (it '
                        'carries its primitives); got %s without primitives' % type(obj).__name__)


def mesh_batches(obj):
    """Triangle meshes behind `obj` (one mesh or the per-sample list of train_sphere.py:58-59,128) as batches of one
    topology: [(sample indices, verts [b,P,3], faces [F,3])].  Meshes that share their face tensor (or equal faces and
    vertex count) render in one launch; the reference renders them one by one (silhouette.py:16-18)."""
    meshes = list(obj) if isinstance(obj, (list, tuple)) else [obj]
    if not meshes:
        raise TypeError('empty list of meshes')
    for m in meshes:
        if not (hasattr(m, 'vertices') and hasattr(m, 'faces')):
            raise TypeError('cannot render %s: neither primitives nor vertices / faces' % type(m).__name__)
    # topologies are compared through host-side fingerprints (ops.faces_fingerprint): the B separately loaded spheres of
    # train_sphere.py:58-59 each own a face tensor, and a device torch.equal per mesh would be B - 1 synchronisations
    groups = {}
    for i, m in enumerate(meshes):
        fp = getattr(m, 'faces_key', None)
        key = (tuple(m.vertices.shape), fp() if callable(fp) else faces_fingerprint(m.faces))
        groups.setdefault(key, []).append(i)
    return [(g, torch.stack([meshes[i].vertices for i in g]), meshes[g[0]].faces) for g in groups.values()]
