"""The reference's training step (train.py:243-262) through the drop-in surface: stand-in network -> head
post-processing -> sampler -> Chamfer (view and object centred) + silhouette + VP-diversity + EMD losses ->
backward -> optimizer.  Checks that every op composes under autograd, that gradients reach every network
parameter, and that a few Adam steps reduce the total loss."""
import os
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, 'examples'))


@pytest.mark.gpu
def test_training_step_composes_and_learns():
    import train_step as T
    dev = torch.device('cuda')
    torch.manual_seed(7)
    B, K, n, size = 4, 8, 128, 64                       # K * n = 1024 points on both sides (EMD needs n == m)
    batch = T.make_batch(B, K, n, size, dev, seed=3)
    net = T.Heads(64, K).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    w = (1.0, 1.0, 1.0, 0.1, 1.0)
    history = []
    for it in range(12):
        opt.zero_grad()
        total, parts = T.training_losses(net, *batch, n, w, seed=500)      # same draws: a deterministic objective
        total.backward()
        if it == 0:
            for name, p in net.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, name
            assert all(bool(torch.isfinite(v)) for v in parts.values())
        opt.step()
        history.append(float(total.detach()))
    assert history[-1] < 0.9 * history[0], history


@pytest.mark.gpu
def test_fused_training_step_equals_the_composition_and_learns():
    """The same loop on TrainStepLossFunction (the whole train.py:243-262 loss as one autograd node, the auction on a second
    stream): at the reference's default shape (K = 16, n = 128, B = 8 -> 4 here, 128 x 128 -> 64 x 64) its total and its
    gradient on the network's parameters equal the module composition's, and Adam steps reduce it."""
    import train_step as T
    dev = torch.device('cuda')
    torch.manual_seed(7)
    B, K, n, size = 4, 16, 128, 64
    batch = T.make_batch(B, K, n, size, dev, seed=3)
    net = T.Heads(64, K).to(dev)
    w = (1.0, 1.0, 1.0, 0.1, 1.0)
    grads = {}
    totals = {}
    for name, fn in (('modules', T.training_losses), ('fused', T.training_losses_fused)):
        net.zero_grad()
        total, parts = fn(net, *batch, n, w, seed=500)
        total.backward()
        totals[name] = (float(total.detach()), {k: float(v.detach()) for k, v in parts.items()})
        grads[name] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
    assert abs(totals['fused'][0] - totals['modules'][0]) <= 1e-5 * abs(totals['modules'][0]), totals
    rel = float((grads['fused'] - grads['modules']).norm() / grads['modules'].norm())
    assert rel <= 1e-4, rel                                     # north_star's tolerance
    for k, wk in zip(('view_cd', 'obj_cd', 'sil', 'vp_div', 'emd'), w):      # the node returns the WEIGHTED terms
        assert abs(totals['fused'][1][k] - wk * totals['modules'][1][k]) <= 1e-5 * max(1e-3, abs(wk * totals['modules'][1][k])), k
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    history = []
    for it in range(12):
        opt.zero_grad()
        total, _ = T.training_losses_fused(net, *batch, n, w, seed=500)
        total.backward()
        opt.step()
        history.append(float(total.detach()))
    assert history[-1] < 0.9 * history[0], history


@pytest.mark.gpu
def test_train_sphere_step_composes_and_learns():
    """train_sphere.py:104-134 through the drop-in surface (examples/train_sphere_step.py): template meshes deformed in
    place by a stand-in network, sampled, Chamfer + silhouette losses on the TRIANGLE path, backward, Adam: gradients
    reach every parameter and a few steps reduce the loss.  BASELINE config C1 sizes (batch 4, 64 x 64)."""
    import train_sphere_step as T
    dev = torch.device('cuda')
    torch.manual_seed(11)
    import vpn_amd
    vpn_amd.TriangleMesh._calls = 0                     # the sample draws are keyed by a per-process call counter
    batch = T.make_batch(4, 2048, 64, dev, seed=5)
    net = T.Offsets(64, 288).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    history = []
    for it in range(15):
        opt.zero_grad()
        total, parts = T.training_losses(net, *batch, 1024, 1.0)
        total.backward()
        if it == 0:
            for name, p in net.named_parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0, name
        opt.step()
        history.append(float(total.detach()))
    assert history[-1] < 0.9 * history[0], history
