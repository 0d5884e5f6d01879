"""Row f2 — primitive -> mesh adapter (modules/meshing of the reference).  The arithmetic is
transform_points(template * v, q, t) (sphere.py:15-22), whose oracle is pinned by the g6 golden vectors; kaolin's
TriangleMesh and the OBJ assets are not used (a procedural 128-vertex template by default)."""
import pytest
import torch

from conftest import rel_err
from oracle import vpn_oracle as O

DEV = 'cuda'


def test_templates_and_obj_reader(tmp_path):
    from vpn_amd.modules import meshing as M
    v, f = M.uv_sphere()
    assert v.shape == (128, 3) and f.dtype == torch.int64 and int(f.max()) == 127 and int(f.min()) == 0
    assert abs(float(v.norm(dim=1).mean()) - 1.0) < 1e-6 and float(v.mean(0).abs().max()) < 1e-6    # sphere.py:33-34
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(1).values
    assert int(torch.unique(e, dim=0, return_counts=True)[1].max()) == 2                             # closed surface
    bv, bf = M.unit_box()
    assert bv.shape == (150, 3) and float(bv.abs().max()) == 1.0 and int(bf.max()) == 149
    p = tmp_path / 't.obj'
    p.write_text('# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n')
    ov, of = M.load_obj(str(p))
    assert ov.shape == (4, 3) and of.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]
    m = M.Meshing.compose_meshes([M.TriangleMesh(ov, of), M.TriangleMesh(ov + 1, of)])
    assert m.vertices.shape == (8, 3) and m.faces[3:].min() == 4                                    # meshing.py:38-39
    with pytest.raises(AssertionError):
        M.Meshing.check_parameters(torch.rand(2, 3), torch.rand(2, 3), torch.rand(2, 3))
    import vpn_amd._lib as lib
    assert lib.lib().vpn_mesh_fwd(None, None, None, None, None, 1, 1, 8, None, None) == -1


@pytest.mark.gpu
def test_mesh_primitives_vs_oracle():
    import vpn_amd
    from vpn_amd.modules import meshing as M
    gen = torch.Generator().manual_seed(5)
    B, kinds = 3, [1, 0, 0, 1, 0]
    K = len(kinds)
    v = (torch.rand(B, K, 3, generator=gen) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    params = torch.cat([v, torch.rand(B, K, 4, generator=gen), 0.35 * (torch.rand(B, K, 3, generator=gen) * 2 - 1)], 2)
    pc = params.clone().requires_grad_(True)
    tpl = {0: M.uv_sphere(), 1: M.unit_box()}
    ref = torch.cat([O.transform_points(tpl[k][0][None] * pc[:, i, None, 0:3], pc[:, i, 3:7], pc[:, i, 7:10])
                     for i, k in enumerate(kinds)], 1)                       # sphere.py:15-22 per primitive, then cat
    W = torch.randn(ref.shape, generator=gen)
    (ref * W).sum().backward()
    pg = params.to(DEV).requires_grad_(True)
    verts, faces = vpn_amd.Meshing.mesh_primitives(pg, kinds)
    (verts * W.to(DEV)).sum().backward()
    assert verts.shape == ref.shape and faces.shape == (2 * 192 + 3 * 252, 3)      # 252 faces: the count of the reference sphere template
    assert int(faces.max()) == verts.shape[1] - 1
    assert rel_err(verts.detach().cpu(), ref.detach()) <= 1e-5
    assert rel_err(pg.grad.cpu(), pc.grad) <= 1e-5
    # the reference's per-primitive surface: a list over the batch, composed per sample
    i = 2
    meshes = vpn_amd.Meshing.sphere_meshing(pg.detach()[:, i, 0:3], pg.detach()[:, i, 3:7], pg.detach()[:, i, 7:10])
    assert len(meshes) == B and meshes[0].vertices.shape == (128, 3)
    o = 150 + 128
    assert torch.equal(meshes[1].vertices, verts.detach()[1, o:o + 128])
    whole = vpn_amd.Meshing.compose_meshes([vpn_amd.Meshing.cuboid_meshing(pg.detach()[:, 0, 0:3], pg.detach()[:, 0, 3:7],
                                                                            pg.detach()[:, 0, 7:10])[0], meshes[0]])
    assert whole.vertices.shape == (150 + 128, 3) and int(whole.faces.max()) == 150 + 127
