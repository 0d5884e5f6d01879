"""CPU replay of the auction on the C5 step clouds (bench.synth_inputs + the oracle sampler); DESIGN.md 4.4 quotes its output
(profiles/r04b_emd_price_sim.txt).  python tools/emd_price_sim.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vpn_oracle as O
import bench
B,K,n=1,64,32
params, gt = bench.synth_inputs(B,K,K*n,1234,'cpu')
u = O.philox_uniforms(1234, 0, B, K, n)
a = O.sample_primitives(params, [0]*K, u)[0].numpy().astype(np.float32); c = gt[0].numpy().astype(np.float32)
N=a.shape[0]
mn=c.min(0); ext=c.max(0)-mn
def cells(g):
    cw=ext/np.array(g); ce=np.minimum(np.floor((c-mn)/cw).astype(int), np.array(g)-1)
    return cw, ce[:,0]*g[1]*g[2]+ce[:,1]*g[2]+ce[:,2], ce
grids={'32x8x8':(32,8,8),'8x8x8':(8,8,8),'16x8x8':(16,8,8), '4x4x4':(4,4,4)}
assign=np.full(N,-1); inv=np.full(N,-1); price=np.zeros(N,np.float32)
memo={}
tot={k:0 for k in grids}; tot_ball=0; tot_need=0; tot_bids=0; tot_ball_cells=0
for it in range(50):
    U=np.nonzero(assign==-1)[0]
    d=np.sqrt(((c[None,:,:]-a[U,None,:])**2).sum(-1)).astype(np.float32)
    val=(3-d)-price[None,:]
    bi=val.argmax(1); r=np.arange(U.size); bv=val[r,bi]; v2=val.copy(); v2[r,bi]=-np.inf; si=v2.argmax(1); sv=v2[r,si]
    R=np.empty(U.size,np.float32)
    for k,i in enumerate(U):
        if i in memo:
            t1,t2=memo[i]; R[k]=3-min(val[k,t1],val[k,t2])+1e-5
        else: R[k]=3-sv[k]+1e-5
    ball=(d<=R[:,None]); need=(d+price[None,:]<=R[:,None])
    tot_ball+=ball.sum(); tot_need+=need.sum(); tot_bids+=U.size
    for name,g in grids.items():
        cw,cid,ce=cells(g)
        ncell=g[0]*g[1]*g[2]
        mp=np.full(ncell,np.inf,np.float32); np.minimum.at(mp,cid,price)
        # d_min(bidder, cell box)
        lo=mn+ce*cw; hi=lo+cw                       # per target: its cell's box
        x=a[U][:,None,:]
        dd=np.maximum(np.maximum(lo[None]-x, x-hi[None]),0); dmin=np.sqrt((dd**2).sum(-1))
        keep=(dmin+mp[cid][None,:]<=R[:,None])
        keep0=(dmin<=R[:,None])
        tot[name]+=keep.sum()
        if name=='32x8x8': tot_ball_cells+=keep0.sum()
    if it in (5,10,20,35,49):
        print('round %d U=%d  per bid: ball %.0f  cells-of-ball(32x8x8) %.0f  needed(d+price<=R) %.0f | kept with cell min price: %s   price median %.3f max %.3f  R median %.3f'%(
            it,U.size,tot_ball/tot_bids,tot_ball_cells/tot_bids,tot_need/tot_bids,' '.join('%s %.0f'%(k,v/tot_bids) for k,v in tot.items()), np.median(price), price.max(), np.median(R)))
    for k,i in enumerate(U): memo[i]=(bi[k],si[k])
    if it==49: break
    inc=(bv-sv)+np.float32(0.005)
    order=np.lexsort((U,-inc.astype(np.float64),bi)); first=np.unique(bi[order],return_index=True)[1]; win=order[first]
    wi,wt=U[win],bi[win]; prev=inv[wt]; assign[prev[prev!=-1]]=-1; inv[wt]=wi; assign[wi]=wt; price[wt]+=inc[win]
