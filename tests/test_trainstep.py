"""BASELINE config C5, single-GPU part: the whole training step of the reference (train.py:243-262) as one autograd node
(TrainStepLossFunction) against (a) the CPU oracle term by term -- sampler (Philox replay), view-centred and object-centred
Chamfer, raster + L1 silhouette, VP-diversity, the auction's EMD -- and (b) the same step composed of the drop-in modules
the way train.py composes it."""
import pytest
import torch

from oracle import vpn_oracle as O
from conftest import ROOT  # noqa: F401

DEV = 'cuda'


def _batch(B, K, n, H, seed, dev):
    g = torch.Generator().manual_seed(seed)
    v = (torch.rand(B, K, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    params = torch.cat([v, torch.rand(B, K, 4, generator=g), 0.35 * (torch.rand(B, K, 3, generator=g) * 2 - 1)], 2)
    gt_view = torch.rand(B, K * n, 3, generator=g) - 0.5
    dists = 1.0 + 0.5 * torch.rand(B, generator=g)
    elevs = 20.0 + 20.0 * torch.rand(B, generator=g)
    azims = 360.0 * torch.rand(B, generator=g)
    angles = 30.0 * torch.rand(B, generator=g)
    gt_canon = O.view_to_obj_points(gt_view, dists, elevs, azims, angles)
    gt_sil = (torch.rand(B, 1, H, H, generator=g) > 0.6).float()
    return [x.to(dev) for x in (params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles)]


def _oracle_step(params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, K, n, H, w, seed):
    return O.train_step(params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, [0] * K, n, H, H, w, seed)


@pytest.mark.gpu
@pytest.mark.parametrize('w', [(1.0, 0.0, 1.0, 0.1, 1.0), (1.0, 0.7, 1.0, 0.1, 1.0), (1.0, 0.5, 0.0, 0.1, 0.0)])
def test_trainstep_node_vs_oracle(w):
    import vpn_amd
    B, K, n, H = 3, 16, 32, 64                     # N = M = 512 points (the auction needs equal sizes, emd_module.py:36; the fused path N M >= 512^2)
    params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles = _batch(B, K, n, H, 5, DEV)
    kinds = vpn_amd.kinds_tensor([0] * K, DEV)
    if not vpn_amd._lib.lib().vpn_hotpath_fused_features(B, K, n, K * n):
        pytest.skip('shape outside the fused path')
    pg = params.clone().requires_grad_(True)
    out = vpn_amd.TrainStepLossFunction.apply(pg, kinds, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, n, 77, 0, H, H, w)
    out[5].backward()
    ref, gref = _oracle_step(*[x.cpu() for x in (params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles)], K, n, H, w, 77)
    got = torch.stack([o.detach() for o in out]).cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6), (got, ref)           # north_star: 1e-4 relative fp32
    err = float((pg.grad.cpu() - gref).abs().max() / gref.abs().max())
    assert err <= 1e-4, err
    assert not any(o.requires_grad for o in out[:5]) and out[5].requires_grad


@pytest.mark.gpu
def test_trainstep_node_equals_the_module_composition():
    """The same step written the way train.py writes it, one drop-in module per reference call: same terms, same gradient
    (the EMD assignment and both Chamfer argmins are bit-equal by the kernels' contracts, so only summation order differs)."""
    import vpn_amd
    B, K, n, H = 4, 16, 32, 64
    params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles = _batch(B, K, n, H, 9, DEV)
    kinds = vpn_amd.kinds_tensor([0] * K, DEV)
    if not vpn_amd._lib.lib().vpn_hotpath_fused_features(B, K, n, K * n):
        pytest.skip('shape outside the fused path')
    w = (1.0, 0.3, 1.0, 0.1, 1.0)
    pa = params.clone().requires_grad_(True)
    out = vpn_amd.TrainStepLossFunction.apply(pa, kinds, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, n, 123, 0, H, H, w)
    out[5].backward()
    pb = params.clone().requires_grad_(True)
    _, _, translates = vpn_amd.split_primitives(pb)
    pred = vpn_amd.Sampling.sample_primitives(pb, kinds, n, seed=123)
    cd = vpn_amd.ChamferDistanceLoss()
    ones, zeros = torch.ones(B, device=DEV), torch.zeros(B, device=DEV)
    terms = [cd(pred, gt_view) * w[0],
             cd(vpn_amd.view_to_obj_points(pred, dists, elevs, azims, angles), gt_canon) * w[1],
             vpn_amd.SilhouetteLoss()(vpn_amd.PrimitivePack(pb, kinds), gt_sil, ones, zeros, zeros) * w[2],
             vpn_amd.VPDiverseLoss(vp_num=K)(translates, gt_view) * w[3],
             torch.sqrt(vpn_amd.EarthMoverDistanceLoss()(pred, gt_view, 0.005, 50)[0]).mean() * w[4]]
    total = sum(terms)
    total.backward()
    for a, b_ in zip(out[:5], terms):
        assert abs(float(a) - float(b_.detach())) <= 2e-6 * max(1.0, abs(float(b_.detach()))), (float(a), float(b_.detach()))
    assert abs(float(out[5].detach()) - float(total.detach())) <= 2e-6 * abs(float(total.detach()))
    err = float((pa.grad - pb.grad).abs().max() / pb.grad.abs().max())
    assert err <= 2e-5, err


@pytest.mark.gpu
def test_trainstep_node_at_config5_shape_replays_in_a_graph():
    """K = 64, n = 32 (N = M = 2048), 256 x 256: captured once, replayed with fresh Philox draws (device step counter advanced by
    the raster launch), finite losses and gradients, the counter moves by one per replay."""
    import vpn_amd
    B, K, n, H = 8, 64, 32, 256
    params, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles = _batch(B, K, n, H, 3, DEV)
    kinds = vpn_amd.kinds_tensor([0] * K, DEV)
    w = (1.0, 0.0, 1.0, 0.1, 1.0)
    p = params.clone().requires_grad_(True)
    seed = torch.full((1,), 1000, dtype=torch.int64, device=DEV)
    one = torch.ones((), device=DEV)

    def step():
        p.grad = None
        out = vpn_amd.TrainStepLossFunction.apply(p, kinds, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, n, seed, 0, H, H, w,
                                                  0.005, 50, True)
        out[5].backward(one)
        return out
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    s0 = int(seed.item())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    totals = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        totals.append(float(out[5]))
        assert bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0
    assert int(seed.item()) == s0 + 3
    assert len(set(totals)) == 3, totals             # fresh surface points every replay
