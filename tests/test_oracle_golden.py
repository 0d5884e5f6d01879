"""Pins oracle/vpn_oracle.py against golden vectors captured from the reference
(oracle/make_golden.py).  CPU only."""
import torch

from oracle import vpn_oracle as O
from conftest import load_golden, rel_err

TOL = 1e-6   # oracle-vs-reference on CPU: same ATen ops, only association order may differ


def _sampler_grads(fn, g):
    v = g['v'].clone().requires_grad_(True)
    q = g['q'].clone().requires_grad_(True)
    t = g['t'].clone().requires_grad_(True)
    pts = fn(v, q, t)
    (pts * g['W']).sum().backward()
    return pts.detach(), v.grad, q.grad, t.grad


def test_g1_sphere_sampler():
    for name in ('g1_sphere_b4_n128', 'g1_sphere_b2_n7'):
        g = load_golden(name)
        pts, gv, gq, gt = _sampler_grads(lambda v, q, t: O.sphere_sampling(v, q, t, g['u'][..., 0], g['u'][..., 1]), g)
        assert rel_err(pts, g['points']) <= TOL
        assert rel_err(gv, g['grad_v']) <= 1e-5
        assert rel_err(gq, g['grad_q']) <= 1e-5
        assert rel_err(gt, g['grad_t']) <= 1e-5


def test_g2_cuboid_sampler():
    g = load_golden('g2_cuboid_b3_n128')
    assert torch.equal(O.cuboid_face_counts(g['v'], 128), g['counts'])
    assert int(g['counts'].sum(1).min()) == 128
    pts, gv, gq, gt = _sampler_grads(lambda v, q, t: O.cuboid_sampling(v, q, t, g['u']), g)
    assert rel_err(pts, g['points']) <= TOL
    assert rel_err(gv, g['grad_v']) <= 1e-5
    assert rel_err(gq, g['grad_q']) <= 1e-5
    assert rel_err(gt, g['grad_t']) <= 1e-5


def test_g3_multi_primitive_order():
    g = load_golden('g3_multi_b2_k3_n16')
    pts = O.sample_primitives(g['params'], g['types'].tolist(), g['u'])
    assert pts.shape == g['points'].shape
    assert rel_err(pts, g['points']) <= TOL


def test_g4_chamfer():
    for name in ('g4_chamfer_b4_n128_m96', 'g4_chamfer_b2_n257_m2048', 'g4_chamfer_ties'):
        g = load_golden(name)
        p1 = g['p1'].clone().requires_grad_(True)
        p2 = g['p2'].clone().requires_grad_(True)
        m1, i1, m2, i2 = O.chamfer_nn(p1, p2)
        assert torch.equal(i1.int(), g['idx1']) and torch.equal(i2.int(), g['idx2'])
        assert torch.equal(m1.detach(), g['min1']) and torch.equal(m2.detach(), g['min2'])
        loss = O.chamfer_loss(p1, p2)
        assert rel_err(loss.detach(), g['loss']) <= TOL
        loss.backward()
        # coincident pair -> NaN gradient in the reference (0/0): same NaN pattern, same finite values
        for mine, ref in ((p1.grad, g['grad_p1']), (p2.grad, g['grad_p2'])):
            assert torch.equal(torch.isnan(mine), torch.isnan(ref))
            ok = ~torch.isnan(ref)
            assert rel_err(mine[ok], ref[ok]) <= 1e-5
        if 'loss_each_w' in g:
            lb = O.chamfer_loss(g['p1'], g['p2'], each_batch=True, w1=0.5, w2=2.0)
            assert rel_err(lb, g['loss_each_w']) <= TOL
            lc = O.chamfer_loss_chunked(g['p1'], g['p2'], w1=0.5, w2=2.0, chunk=1)
            assert rel_err(lc, g['loss_each_w']) <= TOL


def test_g4_tie_and_nan_semantics():
    g = load_golden('g4_chamfer_ties')
    # duplicated target rows 0 and 2: the first index wins
    assert g['idx1'][0, 2].item() == 0
    assert torch.isnan(g['grad_p1']).any()


def test_g5_vpdiverse():
    g = load_golden('g5_vpdiverse_b3_k16_m64')
    ts = [g['translates'][:, k].clone().requires_grad_(True) for k in range(g['translates'].shape[1])]
    loss = O.vp_diverse_loss(ts, g['gt'])
    assert rel_err(loss.detach(), g['loss']) <= TOL
    loss.backward()
    assert rel_err(torch.stack([t.grad for t in ts], 1), g['grad_t']) <= 1e-5


def test_g6_transforms():
    g = load_golden('g6_transforms')
    assert rel_err(O.obj_to_view_points(g['points'], g['dists'], g['elevs'], g['azims']), g['obj_to_view']) <= TOL
    assert rel_err(O.view_to_obj_points(g['points'], g['dists'], g['elevs'], g['azims'], g['angles']),
                   g['view_to_obj']) <= TOL
    assert rel_err(O.rotate_points_forward_x_axis(g['points'], g['angles']), g['rot_x']) <= TOL
    # round trip (angle 0): view_to_obj(obj_to_view(p)) == p
    rt = O.view_to_obj_points(O.obj_to_view_points(g['points'], g['dists'], g['elevs'], g['azims']),
                              g['dists'], g['elevs'], g['azims'], torch.zeros(3))
    assert rel_err(rt, g['points']) <= 1e-5
    p = g['points'].clone().requires_grad_(True)
    q = g['q'].clone().requires_grad_(True)
    t = g['t'].clone().requires_grad_(True)
    out = O.transform_points(p, q, t)
    assert rel_err(out.detach(), g['transform']) <= TOL
    (out * g['W']).sum().backward()
    assert rel_err(p.grad, g['grad_points']) <= 1e-5
    assert rel_err(q.grad, g['grad_q']) <= 1e-5
    assert rel_err(t.grad, g['grad_t']) <= 1e-5


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        r = O.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))
        assert tuple(int(x) for x in r[0]) == out
    u = O.philox_uniforms(1234, 0, 2, 3, 5)
    assert u.shape == (2, 3, 5, 3) and float(u.min()) >= 0 and float(u.max()) < 1
    # sharding invariance: samples [1,2) of a 2-batch == a 1-batch with sample_base=1
    assert torch.equal(u[1:2], O.philox_uniforms(1234, 1, 1, 3, 5))


def test_chamfer_ieee_variant_matches_reference_indices():
    """torch's CPU sqrt is MKL VML (<= 1 ulp, not correctly rounded): the IEEE-sqrt variant of the
    oracle gives the same argmin on every fixture and distances within 1 ulp."""
    for name in ('g4_chamfer_b4_n128_m96', 'g4_chamfer_b2_n257_m2048', 'g4_chamfer_ties'):
        g = load_golden(name)
        m1, i1, m2, i2 = O.chamfer_nn_ieee(g['p1'], g['p2'])
        assert torch.equal(i1.int(), g['idx1']) and torch.equal(i2.int(), g['idx2'])
        for a, b in ((m1, g['min1']), (m2, g['min2'])):
            assert int((a.view(torch.int32) - b.view(torch.int32)).abs().max()) <= 1
