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


@pytest.mark.gpu
def test_reference_mesh_list_call_pattern():
    """train.py:122-149 (get_vp_meshes -> compose_vp_meshes) followed by train.py:165-176 (calculate_silhouette_loss)
    written as the reference writes them: K lists of (B,3|4|3) head outputs -> K x B primitive meshes -> B composed
    meshes -> SilhouetteLoss()(predict_meshes, silhouettes, dists, elevs, azims); and VertexRenderer.render on one
    composed mesh.  Result and gradient equal the PrimitivePack path and the oracle."""
    import vpn_amd
    from vpn_amd import Meshing, SilhouetteLoss, VertexRenderer
    gen = torch.Generator().manual_seed(17)
    BATCH_SIZE, CUBOID_NUM, SPHERE_NUM, CONE_NUM = 3, 2, 3, 0
    K, H, W = CUBOID_NUM + SPHERE_NUM, 48, 48
    v = (torch.rand(BATCH_SIZE, K, 3, generator=gen) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    params = torch.cat([v, torch.rand(BATCH_SIZE, K, 4, generator=gen),
                        0.35 * (torch.rand(BATCH_SIZE, K, 3, generator=gen) * 2 - 1)], 2)
    volumes = [params[:, k, 0:3].to(DEV).requires_grad_(True) for k in range(K)]       # the network's output format
    rotates = [params[:, k, 3:7].to(DEV).requires_grad_(True) for k in range(K)]
    translates = [params[:, k, 7:10].to(DEV).requires_grad_(True) for k in range(K)]
    silhouettes = (torch.rand(BATCH_SIZE, 1, H, W, generator=gen) > 0.5).float().to(DEV)
    dists, elevs, azims = (torch.rand(BATCH_SIZE).to(DEV) for _ in range(3))           # dataset values, overwritten below

    def get_vp_meshes(volumes, rotates, translates):                                   # train.py:122-139
        vp_num = CUBOID_NUM + SPHERE_NUM + CONE_NUM
        meshing_funcs = [Meshing.cuboid_meshing, Meshing.sphere_meshing, Meshing.cone_meshing]
        batch_vp_meshes = [[] for i in range(BATCH_SIZE)]
        meshing_type = 0
        for i in range(vp_num):
            if i == CUBOID_NUM or i == CUBOID_NUM + SPHERE_NUM:
                meshing_type += 1
            meshing = meshing_funcs[meshing_type]
            meshes = meshing(volumes[i], rotates[i], translates[i])
            for b in range(BATCH_SIZE):
                batch_vp_meshes[b].append(meshes[b])
        return batch_vp_meshes

    def compose_vp_meshes(batch_vp_meshes):                                            # train.py:142-149
        batch_meshes = []
        for b in range(len(batch_vp_meshes)):
            batch_meshes.append(Meshing.compose_meshes(batch_vp_meshes[b]))
        return batch_meshes

    predict_meshes = compose_vp_meshes(get_vp_meshes(volumes, rotates, translates))
    assert len(predict_meshes) == BATCH_SIZE and predict_meshes[0].vertices.shape == (2 * 150 + 3 * 128, 3)
    silhouette_loss_func = SilhouetteLoss()                                            # train.py:169
    dists = torch.full_like(dists, fill_value=1.0).to(DEV)                             # train.py:172-174
    elevs, azims = torch.zeros_like(elevs).to(DEV), torch.zeros_like(azims).to(DEV)
    loss = silhouette_loss_func(predict_meshes, silhouettes, dists, elevs, azims)      # train.py:176
    loss.backward()
    # oracle
    kinds = [1] * CUBOID_NUM + [0] * SPHERE_NUM
    pc = params.clone().requires_grad_(True)
    cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(BATCH_SIZE, 3)
    a_ref, _ = O.raster(pc, kinds, cam, H, W, vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA,
                        vpn_amd.config.RASTER_Z_FAR)
    l_ref = O.silhouette_loss(a_ref, silhouettes.cpu())
    l_ref.backward()
    assert rel_err(loss.detach().cpu(), l_ref.detach()) <= 1e-4
    got = torch.cat([torch.stack([x.grad for x in volumes], 1), torch.stack([x.grad for x in rotates], 1),
                     torch.stack([x.grad for x in translates], 1)], 2).cpu()
    assert rel_err(got, pc.grad) <= 1e-4
    # same numbers as the packed path
    l_pack = silhouette_loss_func(vpn_amd.PrimitivePack(params.to(DEV), kinds), silhouettes, dists, elevs, azims)
    assert torch.equal(l_pack, loss.detach())
    # silhouette.py:17: VertexRenderer.render(predict_meshes[i], dists[i], elevs[i], azims[i]) on one composed mesh
    _, alpha, _ = VertexRenderer.render(predict_meshes[1], dists[1], elevs[1], azims[1], image_size=(H, W))
    assert alpha.shape == (1, H, W, 1) and rel_err(alpha[0, :, :, 0].detach().cpu(), a_ref[1].detach()) <= 1e-4
    # a mesh that carries no primitives (loaded from a file, say) is not rendered by the primitive raster
    # (PrimitivePack.of refuses it) but through its triangles (tests/test_mesh_path.py): a finite loss of the same solid
    bare = vpn_amd.TriangleMesh(predict_meshes[0].vertices.detach(), predict_meshes[0].faces)
    with pytest.raises(TypeError):
        vpn_amd.PrimitivePack.of([bare] * BATCH_SIZE)
    l_tri = silhouette_loss_func([bare] * BATCH_SIZE, silhouettes, dists, elevs, azims)
    assert bool(torch.isfinite(l_tri)) and 0.0 < float(l_tri) < 1.0
    mixed = Meshing.compose_meshes([predict_meshes[0], bare])
    assert mixed.primitives is None
