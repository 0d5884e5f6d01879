"""Row f1 — Earth Mover's Distance by auction (modules/loss/emd of the reference).

PARITY UNPINNED against the reference's CUDA extension: it cannot be built here (nvcc) and ships no stored
answers.  What pins the oracle instead: the auction's own guarantee against an exact assignment solver (a
complete assignment is within n*eps of the optimum), a hand-worked case, and the reference's self-check
(test_emd, emd_module.py:81-95: the distance re-derived from the assignment).  The HIP kernel must then equal
the oracle bit for bit — same assignment, same distances — through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import vpn_oracle as O

DEV = 'cuda'


def _clouds(B, n, seed):
    gen = torch.Generator().manual_seed(seed)
    return torch.rand(B, n, 3, generator=gen), torch.rand(B, n, 3, generator=gen)


# ------------------------------------------------------------------ CPU: the oracle itself

def test_oracle_hand_worked_case():
    """Two points, two targets, crossed: round 1 both bid for their own nearest target."""
    x1 = torch.tensor([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]]])
    x2 = torch.tensor([[[0.9, 0.0, 0.0], [0.1, 0.0, 0.0]]])
    dist, assign = O.emd_auction(x1, x2, 0.005, 50)
    assert assign.tolist() == [[1, 0]]
    assert torch.allclose(dist, torch.tensor([[0.01, 0.01]]), atol=1e-7)
    # contested target: both points closest to target 0; the larger increment (point 0) wins, 1 moves on
    x1 = torch.tensor([[[0.0, 0.0, 0.0], [0.2, 0.0, 0.0]]])
    x2 = torch.tensor([[[0.1, 0.0, 0.0], [1.0, 0.0, 0.0]]])
    dist, assign = O.emd_auction(x1, x2, 0.005, 50)
    assert sorted(assign[0].tolist()) == [0, 1]
    assert assign.tolist() == [[0, 1]]                       # 0.1 + 0.8 beats 1.0 + 0.1


def test_oracle_is_eps_optimal_against_exact_solver():
    from scipy.optimize import linear_sum_assignment
    n, eps = 128, 2e-3
    x1, x2 = _clouds(3, n, 1)
    dist, assign = O.emd_auction(x1, x2, eps, 20000)
    for b in range(3):
        assert assign[b].unique().numel() == n, 'auction did not complete; raise iters'
        C = torch.cdist(x1[b].double(), x2[b].double()).numpy()
        r, c = linear_sum_assignment(C)
        opt = C[r, c].sum()
        got = dist[b].double().sqrt().sum().item()
        assert opt - 1e-6 <= got <= opt + n * eps + 1e-4      # Bertsekas: within n*eps of the optimum


def test_oracle_last_iteration_assigns_everyone_and_dist_is_rederivable():
    x1, x2 = _clouds(2, 300, 2)
    for iters in (1, 2, 7):
        dist, assign = O.emd_auction(x1, x2, 0.005, iters)
        assert int(assign.min()) >= 0 and int(assign.max()) < 300
        picked = np.take_along_axis(x2.numpy(), assign.numpy()[..., None].astype(np.int64), axis=1)
        d = ((x1.numpy() - picked) ** 2).sum(-1)              # emd_module.py:91-94
        assert np.allclose(dist.numpy(), d, rtol=1e-6, atol=1e-9)
    # one iteration: everybody gets its nearest target (prices are zero)
    dist, assign = O.emd_auction(x1, x2, 0.005, 1)
    nn = torch.cdist(x1, x2).argmin(-1)
    assert (assign.long() == nn).float().mean() > 0.999


def test_oracle_backward_is_gradient_of_dist_for_fixed_assignment():
    x1, x2 = _clouds(2, 64, 3)
    dist, assign = O.emd_auction(x1, x2, 0.005, 50)
    a = x1.clone().requires_grad_(True)
    picked = torch.gather(x2, 1, assign.long()[..., None].expand(-1, -1, 3))
    g = torch.rand(2, 64)
    (((a - picked) ** 2).sum(-1) * g).sum().backward()
    assert torch.allclose(a.grad, O.emd_backward(x1, x2, g, assign), rtol=1e-6, atol=1e-8)


def test_cabi_validation_and_surface():
    import inspect
    import vpn_amd
    import vpn_amd._lib as lib
    L = lib.lib()
    assert L.vpn_emd_workspace(3, 1000) == 3 * 1000 * 10 * 4 + 24     # 10 words of state per point + (arrival counter, gave-up flag) per sample
    assert L.vpn_emd_workspace(0, 5) == 0
    assert L.vpn_emd_fwd(None, None, 1, 8, 0.005, 50, None, None, None, 0, None) == -1
    assert L.vpn_emd_bwd(None, None, None, None, 1, 8, None, None) == -1
    from vpn_amd.modules import loss
    assert list(inspect.signature(loss.EarthMoverDistanceLoss.forward).parameters) == \
        ['self', 'input1', 'input2', 'eps', 'iters']          # emd_module.py:77
    with pytest.raises(RuntimeError, match='GPU only'):
        loss.EarthMoverDistanceLoss()(torch.rand(1, 8, 3), torch.rand(1, 8, 3), 0.005, 5)
    with pytest.raises(AssertionError):                        # emd_module.py:36
        loss.EarthMoverDistanceLoss()(torch.rand(1, 8, 3), torch.rand(1, 9, 3), 0.005, 5)


# ------------------------------------------------------------------ GPU: kernel == oracle

@pytest.fixture(scope='module')
def emd():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU (no CPU fallback exists)')
    import vpn_amd
    vpn_amd._lib.lib()
    return vpn_amd.modules.loss.EarthMoverDistanceLoss()


def _exact(emd, x1, x2, eps, iters):
    dist, assign = emd(x1.to(DEV), x2.to(DEV), eps, iters)
    torch.cuda.synchronize()
    rd, ra = O.emd_auction(x1, x2, eps, iters)
    assert assign.dtype == torch.int32 and assign.shape == ra.shape
    assert torch.equal(assign.cpu(), ra), 'assignment differs from the oracle at %d places' % int((assign.cpu() != ra).sum())
    assert torch.equal(dist.cpu(), rd)                         # bit-exact
    return dist, assign


@pytest.mark.gpu
@pytest.mark.parametrize('B,n,eps,iters', [
    (3, 64, 0.005, 50), (2, 1, 0.005, 3), (2, 37, 0.01, 200), (2, 1000, 0.005, 50),
    (2, 2048, 0.005, 50),          # train.py:193 at the reference's SAMPLE_NUM * VP_NUM scale
    (1, 1024, 0.002, 1), (1, 1024, 0.002, 2),
    (1, 4500, 0.005, 12),          # more targets than one LDS tile: the cross-tile merge
    (20, 512, 0.005, 50),          # ragged batch (padding workgroups), 8 workgroups per sample
    (70, 300, 0.005, 30),          # more samples than fit with a group: one workgroup per sample
])
def test_emd_equals_oracle(emd, B, n, eps, iters):
    x1, x2 = _clouds(B, n, 100 + n)
    _exact(emd, x1, x2, eps, iters)


@pytest.mark.gpu
def test_emd_both_kernels_agree(emd):
    """n <= 2048 runs the pruned auction with static bidder ownership and the granule exchange (round 4), VPN_EMD_GRID1=1
    round 3's pruned kernel (two box scans per bid, atomic max + counter barrier in memory), VPN_EMD_NOGRID=1 the same
    rounds with the full scan (what 2048 < n <= 4096 uses), VPN_EMD_STREAMING=1 the streaming kernel (state in memory, two
    barriers) that larger clouds use.  All four must equal the oracle bit for bit -- and therefore each other -- for every
    group size."""
    import os
    from vpn_amd.ops import EmdFunction
    x1, x2 = _clouds(5, 700, 21)
    # clustered, anisotropic and degenerate clouds: the pruned scan's grid must not care
    gen = torch.Generator().manual_seed(4)
    y1 = torch.cat([0.05 * torch.randn(2, 300, 3, generator=gen) + 0.5, torch.rand(2, 212, 3, generator=gen)], 1)
    y2 = torch.rand(2, 512, 3, generator=gen) * torch.tensor([1.0, 0.01, 0.0]) + torch.tensor([0.0, 0.3, 0.25])   # a flat sheet
    cases = [(x1, x2, 0.005, 40), (y1, y2, 0.005, 40), (y2, y1, 0.01, 25), (x1[:, :64], x2[:, :64], 0.005, 30)]
    refs = [O.emd_auction(a, b, e, it) for a, b, e, it in cases]
    try:
        for streaming, nogrid, grid1 in (('0', '0', '0'), ('0', '0', '1'), ('0', '1', '0'), ('1', '0', '0')):
            os.environ['VPN_EMD_STREAMING'], os.environ['VPN_EMD_NOGRID'], os.environ['VPN_EMD_GRID1'] = streaming, nogrid, grid1
            for (a, b, e, it), (rd, ra) in zip(cases, refs):
                for G in (None, 1, 2, 4, 8, 16):
                    dist, assign = EmdFunction.apply(a.to(DEV), b.to(DEV), e, it, G)
                    assert torch.equal(assign.cpu(), ra) and torch.equal(dist.cpu(), rd), (streaming, nogrid, grid1, G, a.shape)
    finally:
        for k in ('VPN_EMD_STREAMING', 'VPN_EMD_NOGRID', 'VPN_EMD_GRID1'):
            os.environ.pop(k, None)


@pytest.mark.gpu
def test_emd_ties_and_duplicates(emd):
    """Lattice points (equal values everywhere) and duplicated targets: lowest index wins, second best
    counts duplicates, equal increments resolve to the lowest bidder (also across workgroups of a group)."""
    gen = torch.Generator().manual_seed(5)
    x1 = torch.randint(0, 4, (2, 256, 3), generator=gen).float() / 4
    x2 = torch.randint(0, 4, (2, 256, 3), generator=gen).float() / 4
    _exact(emd, x1, x2, 0.005, 60)
    _exact(emd, x1, x1.clone(), 0.005, 60)
    _exact(emd, x1, x2, 0.0, 40)                                # eps = 0: increments can be exactly zero


@pytest.mark.gpu
def test_emd_gradient(emd):
    x1, x2 = _clouds(2, 512, 7)
    a = x1.to(DEV).requires_grad_(True)
    b = x2.to(DEV).requires_grad_(True)
    dist, assign = emd(a, b, 0.005, 50)
    w = torch.rand(2, 512, generator=torch.Generator().manual_seed(8))
    (dist * w.to(DEV)).sum().backward()
    ref = O.emd_backward(x1, x2, w, assign.cpu())
    assert torch.equal(a.grad.cpu(), ref)
    assert torch.count_nonzero(b.grad) == 0                    # emd_module.py:67: zeros for xyz2
    assert not assign.requires_grad


@pytest.mark.gpu
def test_emd_backward_with_unassigned_points():
    """vpn_emd_bwd through the C ABI with assignment = -1 (what a forward whose group barrier gave up leaves, next
    to a NaN distance): the gradient of such a point is NaN and nothing is read through the index -- not the 12 bytes
    in front of xyz2 (sample 0) nor the previous sample's last point."""
    import vpn_amd
    from vpn_amd import _lib
    B, n = 2, 64
    x1, x2 = _clouds(B, n, 3)
    a, b = x1.to(DEV), x2.to(DEV)
    assign = torch.arange(n, dtype=torch.int32, device=DEV).repeat(B, 1).contiguous()
    assign[0, 0] = -1
    assign[1, 5] = -1
    assign[1, 6] = n + 3                       # out of range the other way
    gd = torch.ones(B, n, device=DEV)
    g1 = torch.zeros(B, n, 3, device=DEV)
    _lib.call('vpn_emd_bwd', _lib.ptr(a), _lib.ptr(b), _lib.ptr(gd), _lib.ptr(assign), B, n, _lib.ptr(g1), _lib.stream())
    torch.cuda.synchronize()
    g1 = g1.cpu()
    bad = torch.zeros(B, n, dtype=torch.bool)
    bad[0, 0] = bad[1, 5] = bad[1, 6] = True
    assert bool(torch.isnan(g1[bad]).all()) and bool(torch.isfinite(g1[~bad]).all())
    ref = 2.0 * (x1 - x2)                       # identity assignment elsewhere
    assert torch.equal(g1[~bad], ref[~bad])


@pytest.mark.gpu
def test_emd_reference_selfcheck_at_full_size(emd):
    """test_emd (emd_module.py:81-95) at its own point count: the distance is the one the assignment implies,
    nearly every target is used, and a long auction completes to within n*eps of ... itself rerun (determinism)."""
    gen = torch.Generator().manual_seed(11)
    x1, x2 = torch.rand(4, 8192, 3, generator=gen).to(DEV), torch.rand(4, 8192, 3, generator=gen).to(DEV)
    dist, assign = emd(x1, x2, 0.05, 3000)
    picked = torch.gather(x2, 1, assign.long()[..., None].expand(-1, -1, 3))
    d = ((x1 - picked) ** 2).sum(-1)
    assert torch.allclose(dist, d, rtol=1e-5, atol=1e-9)
    assert 0 <= int(assign.min()) and int(assign.max()) < 8192
    for b in range(4):
        assert assign[b].unique().numel() >= 8192 * 0.99
    dist2, assign2 = emd(x1, x2, 0.05, 3000)
    assert torch.equal(assign, assign2) and torch.equal(dist, dist2)
    emd_value = dist.sqrt().mean().item()
    assert 0.02 < emd_value < 0.2                               # uniform clouds in the unit cube


@pytest.mark.gpu
def test_emd_group_size_does_not_change_the_result():
    """The workgroups of a sample's group (cooperative launch, inter-workgroup barrier) versus one workgroup per sample
    (max_group = 1, what the op uses when other streams share the GPU): bit-identical assignments and distances."""
    import vpn_amd
    from vpn_amd.ops import EmdFunction
    for B, n, eps, iters in ((8, 2048, 0.005, 50), (3, 1000, 0.01, 30), (1, 4100, 0.005, 20)):
        x1, x2 = _clouds(B, n, 900 + n)
        a, b = x1.to(DEV), x2.to(DEV)
        d0, i0 = EmdFunction.apply(a, b, eps, iters, 0)       # automatic: up to 16 workgroups per sample
        d1, i1 = EmdFunction.apply(a, b, eps, iters, 1)
        d2, i2 = EmdFunction.apply(a, b, eps, iters, 4)
        assert torch.equal(i0, i1) and torch.equal(d0, d1) and torch.equal(i0, i2) and torch.equal(d0, d2)
        assert bool(torch.isfinite(d0).all()) and int(i0.min()) >= 0


@pytest.mark.gpu
def test_emd_cooperative_launch_is_opt_in_and_agrees(emd):
    """Round 4: the group's grid is bounded by the occupancy query and launched plainly; VPN_EMD_COOP_LAUNCH=1 adds the
    runtime's residency check (hipLaunchCooperativeKernel).  Same bits either way, for the team kernel (n <= 2048) and the
    streaming one."""
    import os
    from vpn_amd.ops import EmdFunction
    cases = [_clouds(3, 700, 41) + (0.005, 30), _clouds(1, 4500, 42) + (0.005, 8)]
    refs = [O.emd_auction(a, b, e, it) for a, b, e, it in cases]
    try:
        for coop in ('0', '1'):
            os.environ['VPN_EMD_COOP_LAUNCH'] = coop
            for (a, b, e, it), (rd, ra) in zip(cases, refs):
                dist, assign = EmdFunction.apply(a.to(DEV), b.to(DEV), e, it, None)
                assert torch.equal(assign.cpu(), ra) and torch.equal(dist.cpu(), rd), (coop, a.shape)
    finally:
        os.environ.pop('VPN_EMD_COOP_LAUNCH', None)


@pytest.mark.gpu
def test_emd_balanced_form_agrees(emd):
    """Round 4: rounds with many bidders bid in the balanced form (emd.hip: the rows of all own bidders counting-sorted by
    length and dealt over the lanes, the price filter, best / second by LDS atomic maxima of (value, lowest index)); rounds
    (one to 64 lanes per row, a wave per row of 63 targets and more); the first bids, degenerate clouds and -- when the
    switch is moved -- rounds with little work in the team form (VPN_EMD_FLAT_MIN own bidders, VPN_EMD_FLAT_WORK = own
    bidders x targets per bid of the last balanced round).  FLAT_MIN = 0: teams only; default (1 / 0): balanced from the
    second round on; 64 / 4000: a mix of both.  Same bits as the oracle whichever form bids, for every group size -- uniform clouds, the crowded auction
    of the training step (points on small primitives against a cloud that fills the cube: 500+ bidders in every round),
    lattices (ties everywhere: the index rule lives in the keys), clouds far from the origin, and one workgroup per sample
    (2048 own bidders: several batches per round)."""
    import os
    from vpn_amd.ops import EmdFunction
    gen = torch.Generator().manual_seed(77)
    uni = (torch.rand(2, 2048, 3, generator=gen), torch.rand(2, 2048, 3, generator=gen), 0.005, 50)
    centres = torch.rand(2, 64, 1, 3, generator=gen) * 0.7 + 0.15
    blobs = (centres + 0.04 * torch.randn(2, 64, 32, 3, generator=gen)).reshape(2, 2048, 3)
    crowded = (blobs, torch.rand(2, 2048, 3, generator=gen), 0.005, 50)
    lat1 = torch.randint(0, 6, (1, 1024, 3), generator=gen).float() / 6
    lat2 = torch.randint(0, 6, (1, 1024, 3), generator=gen).float() / 6
    lattice = (lat1, lat2, 0.005, 40)
    far = (blobs[:1] * 50.0 + 1000.0, torch.rand(1, 2048, 3, generator=gen) * 50.0 + 1000.0, 0.05, 30)
    # 1030 points, 16 workgroups per sample: 68 own bidders each -- their 18-byte state ends off a 16-byte boundary, the
    # balanced form's lists behind it must not care
    odd = (blobs[:, :1030].contiguous(), torch.rand(2, 1030, 3, generator=gen), 0.005, 30)
    cases = [uni, crowded, lattice, far, odd]
    refs = [O.emd_auction(a, b, e, it) for a, b, e, it in cases]
    try:
        for flat_min, flat_work in (('0', None), ('64', '4000'), (None, None)):
            for key, v in (('VPN_EMD_FLAT_MIN', flat_min), ('VPN_EMD_FLAT_WORK', flat_work)):
                os.environ.pop(key, None)
                if v is not None:
                    os.environ[key] = v
            for (a, b, e, it), (rd, ra) in zip(cases, refs):
                for G in (None, 1, 4, 16):
                    dist, assign = EmdFunction.apply(a.to(DEV), b.to(DEV), e, it, G)
                    assert torch.equal(assign.cpu(), ra) and torch.equal(dist.cpu(), rd), (flat_min, flat_work, G, a.shape)
    finally:
        os.environ.pop('VPN_EMD_FLAT_MIN', None)
        os.environ.pop('VPN_EMD_FLAT_WORK', None)
