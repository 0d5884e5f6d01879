"""Row f4 — head post-processing into packed primitive parameters (vpnet_one_resnet.py:34-41, :67-85).
Parity unpinned in the strict sense (the reference module needs torchvision); the oracle restates its three lines
with torch's own sigmoid / tanh / clamp."""
import inspect

import pytest
import torch

from conftest import rel_err
from oracle import vpn_oracle as O

DEV = 'cuda'


def _raw(B, K, seed, scale=3.0):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3 * K, generator=gen) * scale, torch.randn(B, 4 * K, generator=gen) * scale,
            torch.randn(B, 3 * K, generator=gen) * scale)


def test_oracle_head_matches_the_reference_lines_on_a_hand_case():
    v = torch.tensor([[0.0, 1.0, -1.0, 2.0, 0.5, -0.5]])                      # K = 2
    q = torch.zeros(1, 8)
    t = torch.tensor([[0.0, 1.0, -1.0, 0.5, 0.25, -2.0]])
    p = O.head_post_process(v, q, t)
    assert p.shape == (1, 2, 10)
    assert torch.allclose(p[0, 0, :3], (torch.sigmoid(v[0, :3]) + 0.1) / torch.tensor([8.0, 10.0, 10.0]))
    assert torch.allclose(p[0, 1, :3], (torch.sigmoid(v[0, 3:]) + 0.1) / torch.tensor([8.0, 10.0, 10.0]))
    assert torch.allclose(p[0, :, 3:7], torch.full((2, 4), 0.5))
    assert torch.allclose(p[0, 1, 7:], torch.tanh(t[0, 3:]))
    pc = O.head_post_process(v, q + 3, t, is_sigmoid=False)
    assert torch.allclose(pc[0, 0, :3], torch.tensor([0.01 + 1e-8, 0.8, 0.01 + 1e-8]) / torch.tensor([8.0, 10.0, 10.0]))
    assert torch.allclose(pc[0, :, 3:7], torch.ones(2, 4)) and float(pc[0, 1, 9]) == -1.0


def test_head_surface_and_validation():
    import vpn_amd
    import vpn_amd._lib as lib
    L = lib.lib()
    assert L.vpn_head_pack_fwd(None, None, None, 1, 1, 1, 0.01, 0.8, 8.0, 10.0, 10.0, None, None) == -1
    assert list(inspect.signature(vpn_amd.pack_head_outputs).parameters)[:3] == ['volumes', 'rotates', 'translates']
    p = torch.rand(2, 3, 10)
    v, q, t = vpn_amd.split_primitives(p)
    assert len(v) == len(q) == len(t) == 3 and v[1].shape == (2, 3) and q[2].shape == (2, 4)
    assert torch.equal(vpn_amd.pack_primitives(v, q, t), p)                   # inverse of the packing
    with pytest.raises(RuntimeError, match='GPU only'):
        vpn_amd.pack_head_outputs(*_raw(2, 4, 0))


@pytest.mark.gpu
@pytest.mark.parametrize('is_sigmoid', [True, False])
@pytest.mark.parametrize('B,K', [(3, 5), (64, 32), (1, 1)])
def test_head_pack_vs_oracle(B, K, is_sigmoid):
    import vpn_amd
    v, q, t = _raw(B, K, 10 * B + K, scale=3.0 if is_sigmoid else 1.0)
    if not is_sigmoid:
        v = v * 0.5 + 0.4                                                     # straddle both clamp limits
    leaves = [x.clone().requires_grad_(True) for x in (v, q, t)]
    ref = O.head_post_process(*leaves, is_sigmoid=is_sigmoid)
    W = torch.randn(B, K, 10, generator=torch.Generator().manual_seed(1))
    (ref * W).sum().backward()
    gl = [x.to(DEV).requires_grad_(True) for x in (v, q, t)]
    out = vpn_amd.pack_head_outputs(*gl, is_sigmoid=is_sigmoid)
    (out * W.to(DEV)).sum().backward()
    assert out.shape == (B, K, 10)
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-5
    for mine, theirs in zip(gl, leaves):
        assert rel_err(mine.grad.cpu(), theirs.grad) <= 1e-5
    # the packed output feeds the sampler as is
    kinds = vpn_amd.kinds_tensor([0] * K, torch.device(DEV))
    pts = vpn_amd.Sampling.sample_primitives(out.detach(), kinds, 8, seed=3)
    assert pts.shape == (B, K * 8, 3) and bool(torch.isfinite(pts).all())


def test_on_disk_formats():
    """split.csv, rendering_metadata.txt and RGBA renderings (dataset.py:98-151), on hand-written samples."""
    from vpn_amd.modules import dataset as D
    csv = ('id,synsetId,subSynsetId,modelId,split\n'
           '1,02691156,02691156,aaa111,train\n2,02691156,02690373,bbb222,val\n3,03001627,03001627,ccc333,test\n'
           '4,03001627,03001627,ddd444,train\nbroken,line\n')
    sp = D.parse_split_csv(csv)
    assert sp['train'] == [('02691156', 'aaa111'), ('03001627', 'ddd444'), ('02691156', 'bbb222')]   # val after train
    assert sp['test'] == [('03001627', 'ccc333')]
    meta = '293.65 26.04 0 0.78 25\n118.4 29.9 0 0.91 25\nnot a camera line\n'
    az, el, di = D.parse_rendering_metadata(meta)
    assert az == [293.65, 118.4] and el == [26.04, 29.9]
    assert abs(di[0] - 0.78 * 1.754) < 1e-12 and abs(di[1] - 0.91 * 1.754) < 1e-12                   # dataset.py:147
    img = torch.rand(4, 8, 8)
    rgb, sil = D.split_rgba(img)
    assert torch.equal(rgb, img[:3]) and torch.equal(sil, img[3:4]) and sil.shape == (1, 8, 8)
    rgbn, _ = D.split_rgba(img, normalize=True)
    assert torch.allclose(rgbn[1], (img[1] - 0.456) / 0.224)
    assert D.split_rgba(torch.rand(2, 4, 8, 8))[1].shape == (2, 1, 8, 8)
