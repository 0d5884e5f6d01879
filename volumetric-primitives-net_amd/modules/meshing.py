"""modules/meshing of the reference (meshing.py, sphere.py, cuboid.py) on the HIP mesh kernel: primitive parameters
-> triangle meshes.  The new raster consumes (v, q, t) directly, so this is a compatibility adapter for callers that
want vertices / faces (visual dumps, the GCN's 16 x 128 vertex input, gcn.py:34-35).

Differences from the reference, all forced: kaolin's TriangleMesh is absent, so a minimal holder with the same two
attributes is provided; the template is parsed once and kept on the device (the reference re-parses the OBJ for every
(sample, primitive): sphere.py:14, :30-36); if no OBJ path is given a procedural template with the reference's vertex
count (128) is used — the reference's own `objects/*.obj` can be passed via `set_templates`."""
import math

import torch

from ..ops import SPHERE, CUBOID, MeshFunction, kinds_tensor
from ..primitives import PrimitivePack


class TriangleMesh:
    """The attributes and methods of kaolin.rep.TriangleMesh that the reference touches: `vertices` / `faces`
    (meshing.py:35-43, vertex_renderer.py:20-21), `from_obj`, `cuda` and `sample` (train_sphere.py:53-54,76), plus
    `primitives`: the PrimitivePack (one sample, [1,K,10] + kinds) the mesh was generated from, attached by
    Meshing.*_meshing and carried through compose_meshes.

    How a mesh renders (SilhouetteLoss / VertexRenderer.render):
      * a mesh that carries primitives AND whose vertices are still the ones Meshing wrote renders through the
        primitive raster, exactly as train.py:122-149,176 builds and passes it;
      * any other mesh -- loaded from an OBJ, built from tensors, or a Meshing-made mesh whose vertices were edited
        afterwards (train_sphere.py:62-68 deforms its meshes in place: `mesh.vertices += offset`) -- renders its
        TRIANGLES through the mesh raster (vpn_mesh_raster_fwd/bwd).  An edit is never silently ignored: the
        primitives are only trusted while (vertices tensor, its version counter) are the ones recorded here."""

    def __init__(self, vertices: torch.Tensor, faces: torch.Tensor, primitives=None):
        self.vertices, self.faces = vertices, faces
        self._primitives = primitives
        # the tensor itself is kept (not its id, which a later tensor may reuse) with the version it had
        self._stamp = (vertices, vertices._version) if primitives is not None else None

    @property
    def primitives(self):
        """The primitives this mesh was made from, or None once its vertices have been replaced or edited in place."""
        if self._primitives is None or self._stamp[0] is not self.vertices or self._stamp[1] != self.vertices._version:
            return None
        return self._primitives

    @classmethod
    def from_tensors(cls, vertices, faces, primitives=None):
        return cls(vertices, faces, primitives)

    @classmethod
    def from_obj(cls, path):
        return cls(*load_obj(path))

    def faces_key(self):
        """Content key of the face tensor (ops.faces_fingerprint): computed on the host tensor when there is one."""
        from ..ops import faces_fingerprint
        return faces_fingerprint(self.faces)

    def to(self, device):
        from ..ops import faces_fingerprint, faces_remember
        moved = self.vertices.to(device)
        keep = self.primitives if moved is self.vertices else None
        faces = self.faces.to(device)
        if faces is not self.faces and not self.faces.is_cuda:
            faces_remember(faces, faces_fingerprint(self.faces))       # hashed on the host: the device copy is never read back
        self.vertices, self.faces = moved, faces
        self._primitives = keep
        self._stamp = (self.vertices, self.vertices._version) if keep is not None else None
        return self

    def cuda(self):
        return self.to('cuda')                                   # train_sphere.py:54

    def sample(self, num_samples: int, seed=None):
        """kaolin's TriangleMesh.sample as train_sphere.py:76 uses it: `num_samples` points drawn uniformly from the
        surface (faces chosen in proportion to their area, uniform inside the face) -> (points [n,3], face_choices
        [n]).  Differentiable w.r.t. the vertices.  Draws come from Philox keyed by (seed, a per-process call counter)
        unless a seed is given."""
        from ..ops import MeshSampleFunction, faces_i32
        if seed is None:
            TriangleMesh._calls += 1
            seed, base = TriangleMesh.sample_seed, TriangleMesh._calls
        else:
            base = 0
        pts, idx = MeshSampleFunction.apply(self.vertices[None], faces_i32(self.faces, self.vertices.device), int(num_samples),
                                            None, int(seed), int(base))
        return pts[0], idx[0].long()

    sample_seed = 1234          # config.py:20 MANUAL_SEED of the reference
    _calls = 0


def load_obj(path):
    """Vertices (P,3) fp32 and triangle faces (F,3) int64 of a Wavefront OBJ (v / f records; polygons are fanned,
    `a/b/c` face tokens keep the vertex index, negative indices count from the end)."""
    vs, fs = [], []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v':
                vs.append([float(x) for x in tok[1:4]])
            elif tok[0] == 'f':
                idx = [int(t.split('/')[0]) for t in tok[1:]]
                idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                for j in range(1, len(idx) - 1):
                    fs.append([idx[0], idx[j], idx[j + 1]])
    return torch.tensor(vs, dtype=torch.float32), torch.tensor(fs, dtype=torch.int64).reshape(-1, 3)


def normalize_sphere_template(vertices):
    """load_cuboid of sphere.py:30-36: zero centre, mean vertex norm 1."""
    v = vertices - vertices.mean(0)
    return v / v.norm(dim=1).mean()


def uv_sphere(rings=8, segments=16):
    """rings x segments vertices on the unit sphere (8 x 16 = 128, the reference template's count), no pole
    vertices; quads between rings split into triangles, the two caps fanned."""
    vs, fs = [], []
    for r in range(rings):
        th = math.pi * (r + 0.5) / rings
        for s in range(segments):
            ph = 2.0 * math.pi * s / segments
            vs.append([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
    for r in range(rings - 1):
        for s in range(segments):
            a, b = r * segments + s, r * segments + (s + 1) % segments
            c, d = a + segments, b + segments
            fs += [[a, c, b], [b, c, d]]
    for s in range(1, segments - 1):
        fs.append([0, s, s + 1])
        base = (rings - 1) * segments
        fs.append([base, base + s + 1, base + s])
    return normalize_sphere_template(torch.tensor(vs, dtype=torch.float32)), torch.tensor(fs, dtype=torch.int64)


def unit_box(n=5):
    """Vertices on the surface of [-1,1]^3: an n x n grid per face without duplicates is not needed by any caller,
    so faces keep their own grids (6 n^2 vertices), two triangles per grid cell."""
    vs, fs = [], []
    lin = [-1.0 + 2.0 * i / (n - 1) for i in range(n)]
    for axis in range(3):
        for sign in (1.0, -1.0):
            base = len(vs)
            for a in lin:
                for b in lin:
                    p = [0.0, 0.0, 0.0]
                    p[axis] = sign
                    p[(axis + 1) % 3] = a
                    p[(axis + 2) % 3] = b
                    vs.append(p)
            for i in range(n - 1):
                for j in range(n - 1):
                    q = base + i * n + j
                    tri = [[q, q + n, q + 1], [q + 1, q + n, q + n + 1]]
                    fs += tri if sign > 0 else [[t[0], t[2], t[1]] for t in tri]
    return torch.tensor(vs, dtype=torch.float32), torch.tensor(fs, dtype=torch.int64)


class Meshing:
    _templates = {}          # (kind, device) -> (vertices, faces) on that device
    _layouts = {}            # (kinds, device) -> (vertex offsets [K+1] int32, composed faces, total vertices)
    _sources = {SPHERE: None, CUBOID: None}

    def __init__(self):
        pass

    @classmethod
    def set_templates(cls, sphere_obj=None, cuboid_obj=None):
        """Use OBJ templates (e.g. the reference's modules/meshing/objects/sphere.obj / cuboid.obj)."""
        cls._sources = {SPHERE: sphere_obj, CUBOID: cuboid_obj}
        cls._templates = {}
        cls._layouts = {}

    @classmethod
    def template(cls, kind, device):
        key = (kind, str(device))
        if key not in cls._templates:
            src = cls._sources[kind]
            if src is not None:
                v, f = load_obj(src)
                if kind == SPHERE:
                    v = normalize_sphere_template(v)            # sphere.py:33-34 (the cuboid template is used as is)
            else:
                v, f = uv_sphere() if kind == SPHERE else unit_box()
            cls._templates[key] = (v.to(device).contiguous(), f.to(device))
        return cls._templates[key]

    @classmethod
    def mesh_primitives(cls, params, kinds):
        """All K primitives of all B samples in one launch: params (B,K,10), kinds list ->
        vertices (B, P_total, 3) (differentiable) and the faces (F_total, 3) of the composed mesh."""
        from ..ops import kinds_host
        kinds = tuple(int(k) for k in (kinds_host(kinds) if isinstance(kinds, torch.Tensor) and kinds.is_cuda else
                                       (kinds.tolist() if isinstance(kinds, torch.Tensor) else kinds)))
        dev = params.device
        tpl = {k: cls.template(k, dev) for k in set(kinds)}
        # vertex offsets and the composed face list depend on the kind list only: built once per (kinds, device) -- a host
        # list turned into a device tensor is a synchronising copy, and this runs every training step
        key = (kinds, str(dev))
        if key not in cls._layouts:
            offsets, faces = [0], []
            for k in kinds:
                v, f = tpl[k]
                faces.append(f + offsets[-1])                   # meshing.py:38-39
                offsets.append(offsets[-1] + v.shape[0])
            cls._layouts[key] = (torch.tensor(offsets, dtype=torch.int32, device=dev), torch.cat(faces), offsets[-1])
        off, faces, ptot = cls._layouts[key]
        verts = MeshFunction.apply(params, kinds_tensor(kinds, dev), off,
                                   tpl[SPHERE][0] if SPHERE in tpl else None, tpl[CUBOID][0] if CUBOID in tpl else None, ptot)
        return verts, faces

    @classmethod
    def _one_kind(cls, v, q, t, kind):
        cls.check_parameters(v, q, t)
        params = torch.cat([v, q, t], 1)[:, None, :]
        verts, faces = cls.mesh_primitives(params, [kind])
        kt = kinds_tensor([kind], params.device)
        # a list over the batch (sphere.py:24-27); each mesh remembers the primitive it came from
        return [TriangleMesh(verts[b], faces, PrimitivePack(params[b], kt)) for b in range(v.size(0))]

    @classmethod
    def cuboid_meshing(cls, v: torch.Tensor, q: torch.Tensor, t: torch.Tensor) -> list:
        return cls._one_kind(v, q, t, CUBOID)

    @classmethod
    def sphere_meshing(cls, v: torch.Tensor, q: torch.Tensor, t: torch.Tensor) -> list:
        return cls._one_kind(v, q, t, SPHERE)

    @classmethod
    def cone_meshing(cls, v: torch.Tensor, q: torch.Tensor, t: torch.Tensor) -> list:
        cls.check_parameters(v, q, t)                           # meshing.py:22-25 is `pass` too
        return None

    @staticmethod
    def compose_meshes(meshes: list) -> TriangleMesh:
        """meshing.py:27-46: concatenate vertices, offset faces."""
        vertices, faces, n = [], [], 0
        for m in meshes:
            vertices.append(m.vertices)
            faces.append(m.faces + n)
            n += m.vertices.size(0)
        packs = [getattr(m, 'primitives', None) for m in meshes]
        prims = None
        if all(p is not None for p in packs):                   # K single-primitive packs -> one [1,K,10] pack
            from ..ops import kinds_host
            kinds = [k for p in packs for k in kinds_host(p.kinds)]          # host tuples: no device round trip per mesh
            prims = PrimitivePack(torch.cat([p.params for p in packs], 1), kinds)
        return TriangleMesh.from_tensors(vertices=torch.cat(vertices), faces=torch.cat(faces), primitives=prims)

    @staticmethod
    def check_parameters(v: torch.Tensor, q: torch.Tensor, t: torch.Tensor):
        assert v.size(0) == q.size(0) == t.size(0)              # meshing.py:49-55
        B = v.size(0)
        assert v.size() == (B, 3)
        assert q.size() == (B, 4)
        assert t.size() == (B, 3)
