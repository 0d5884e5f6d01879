"""How many (workgroup, tile) pairs could the Chamfer scan skip at C3?  CPU simulation on the step's clouds with the GT
cloud in Morton order (8 octant-sized tiles of 256), exact distances, a tile skipped when its box lies beyond the current
worst best of the query group; query groups of 256 (the scan's workgroup: its LDS tiles are shared) and 32 (a wave),
tiles in natural and in nearest-first order:   python tools/scan_skip_sim.py > profiles/r04_scan_skip_sim.txt"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vpn_oracle as O
import bench
B,K,n,M=4,32,256,2048
params, gt = bench.synth_inputs(B,K,M,1234,'cpu')
u = O.philox_uniforms(1234, 0, B, K, n)
pts = O.sample_primitives(params, [0]*K, u).numpy()
gt = gt.numpy()
def morton(p, bits=3):
    q = np.clip(((p+0.5)*(1<<bits)).astype(int),0,(1<<bits)-1)
    code = np.zeros(len(p),int)
    for b in range(bits):
        for a in range(3): code |= ((q[:,a]>>b)&1) << (3*b+a)
    return code
def sim(Q, T, qgroup, tile=256, order='natural'):
    # Q: queries [nq,3] grouped in consecutive groups of qgroup; T: targets [nt,3] in tiles of `tile`
    nt=len(T); tiles=[T[i:i+tile] for i in range(0,nt,tile)]
    tlo=np.array([t.min(0) for t in tiles]); thi=np.array([t.max(0) for t in tiles])
    scanned=0; total=0
    for g0 in range(0,len(Q),qgroup):
        q=Q[g0:g0+qgroup]; qlo=q.min(0); qhi=q.max(0)
        gap=np.maximum(0, np.maximum(tlo-qhi, qlo-thi)); bd=np.sqrt((gap**2).sum(1))
        idxs=np.argsort(bd) if order=='nearest' else np.arange(len(tiles))
        best=np.full(len(q),np.inf)
        for ti in idxs:
            total+=1
            if bd[ti] > best.max(): continue
            scanned+=1
            d=np.sqrt(((q[:,None,:]-tiles[ti][None,:,:])**2).sum(-1)).min(1)
            best=np.minimum(best,d)
    return scanned/total
for b in range(2):
    g=gt[b]; gs=g[np.argsort(morton(g,3),kind='stable')]
    p=pts[b]
    for order in ('natural','nearest'):
        for qg in (256, 32):
            d1=sim(p, gs, qg, order=order)          # direction 1: queries = sampled (grouped by primitive), targets = sorted GT
            d2=sim(gs, p, qg, order=order)          # direction 2: queries = sorted GT, targets = sampled (tile = primitive)
            print('sample',b,order,'query group',qg,'dir1 scanned %.2f  dir2 scanned %.2f'%(d1,d2))
