"""One-off stress of the auction against the oracle: random sizes, batch sizes, cloud kinds, eps, rounds and group sizes.
    python tools/emd_stress.py [cases] [seed]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd.ops import EmdFunction
from oracle import vpn_oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device('cuda')
bad = 0
for c in range(cases):
    n = rng.choice([128, 130, 200, 511, 640, 1000, 1024, 1030, 1500, 2047, 2048])
    B = rng.choice([1, 2, 3, 5])
    kind = rng.choice(['uniform', 'blobs', 'lattice', 'sheet', 'far', 'dup'])
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    a, b = torch.rand(B, n, 3, generator=g), torch.rand(B, n, 3, generator=g)
    if kind == 'blobs':
        k = rng.choice([4, 16, 64])
        cen = torch.rand(B, k, 1, 3, generator=g)
        a = (cen + 0.03 * torch.randn(B, k, (n + k - 1) // k, 3, generator=g)).reshape(B, -1, 3)[:, :n].contiguous()
    elif kind == 'lattice':
        q = rng.choice([3, 5, 8])
        a, b = (a * q).floor() / q, (b * q).floor() / q
    elif kind == 'sheet':
        b = b * torch.tensor([1.0, 0.02, 0.0]) + torch.tensor([0.0, 0.4, 0.3])
    elif kind == 'far':
        a, b = a * 30 - 500.0, b * 30 - 500.0
    elif kind == 'dup':
        b[:, n // 2:] = b[:, :n - n // 2].clone()
        a = b.clone() if rng.random() < 0.5 else a
    eps = rng.choice([0.0, 0.002, 0.005, 0.05]) * (30.0 if kind == 'far' else 1.0)
    iters = rng.choice([1, 2, 7, 30, 50])
    G = rng.choice([None, 1, 2, 4, 8, 16])
    rd, ra = O.emd_auction(a, b, eps, iters)
    d, i = EmdFunction.apply(a.to(dev), b.to(dev), eps, iters, G)
    ok = torch.equal(i.cpu(), ra) and torch.equal(d.cpu(), rd)
    bad += not ok
    print('%2d n=%4d B=%d %-8s eps=%.3g iters=%2d G=%s: %s' % (c, n, B, kind, eps, iters, G, 'ok' if ok else 'DIFFERS'), flush=True)
print('differing cases:', bad)
sys.exit(1 if bad else 0)
