"""The auction's workload regimes (CPU, numpy restatement of the rounds of oracle.vpn_oracle.emd_auction with counters):
    python tools/emd_regimes.py > profiles/r04_emd_regimes.txt
per round the number of bidders U, the mean radius R = 3 - second-best value, the targets inside the ball of radius R
("inR": what a bid has to look at) and inside the box of 8x8x8 cells that covers it ("inbox": what round 3's kernel
scanned), for (a) the clouds the training step hands the auction (bench.py c5_inputs: points on K small primitives
against a GT cloud that fills the cube), (b) two uniform clouds (the reference's test_emd pattern), (c) a nearly converged
prediction (GT + 3 % noise)."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vpn_oracle as O
def synth(B,K,M,seed):
    g = torch.Generator().manual_seed(seed)
    v = (torch.rand(B, K, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    q = torch.rand(B, K, 4, generator=g)
    t = 0.35 * (torch.rand(B, K, 3, generator=g) * 2 - 1)
    gt = torch.rand(B, M, 3, generator=g) - 0.5
    return torch.cat([v,q,t],2), gt
B,K,n=1,64,32
params, gt = synth(B,K,K*n,1234)
u = O.philox_uniforms(1234, 0, B, K, n)
pts = O.sample_primitives(params, [0]*K, u)
def stats(x1, x2, eps=0.005, iters=50, label=''):
    a, c = x1[0].numpy().astype(np.float32), x2[0].numpy().astype(np.float32)
    N=a.shape[0]
    assign=np.full(N,-1); inv=np.full(N,-1); price=np.zeros(N,np.float32)
    tot_scans=0; tot_in_R=0; tot_in_box=0
    mn=c.min(0); ext=c.max(0)-mn; cell=ext/8
    rows=[]
    for it in range(iters):
        U=np.nonzero(assign==-1)[0]
        if U.size==0: break
        d=np.sqrt(((c[None,:,:]-a[U,None,:])**2).sum(-1)).astype(np.float32)
        val=(3-d)-price[None,:]
        bi=val.argmax(1); r=np.arange(U.size); bv=val[r,bi]; v2=val.copy(); v2[r,bi]=-np.inf; sv=v2.max(1)
        R=3-sv
        inR=(d<=R[:,None]).sum(1)
        # box count: cells covering [x-R,x+R] per axis
        lo=np.clip(np.floor((a[U]-R[:,None]-mn)/cell),0,7); hi=np.clip(np.floor((a[U]+R[:,None]-mn)/cell),0,7)
        tc=np.clip(np.floor((c-mn)/cell),0,7)
        inbox=((tc[None,:,:]>=lo[:,None,:])&(tc[None,:,:]<=hi[:,None,:])).all(-1).sum(1)
        tot_scans+=U.size; tot_in_R+=inR.sum(); tot_in_box+=inbox.sum()
        rows.append((it,U.size,float(R.mean()),float(inR.mean()),float(inbox.mean())))
        if it==iters-1: break
        inc=(bv-sv)+np.float32(eps)
        order=np.lexsort((U,-inc.astype(np.float64),bi)); first=np.unique(bi[order],return_index=True)[1]; win=order[first]
        wi,wt=U[win],bi[win]; prev=inv[wt]; assign[prev[prev!=-1]]=-1; inv[wt]=wi; assign[wi]=wt; price[wt]+=inc[win]
    print(label,'scans',tot_scans,'mean inR',tot_in_R/tot_scans,'mean inbox',tot_in_box/tot_scans)
    for r in rows[:6]+rows[10::10]: print('   it %2d U %4d  R %.3f  inR %.0f inbox %.0f'%r)
stats(pts, gt, label='C5 synthetic (pred prims vs uniform gt)')
g=torch.Generator().manual_seed(1); x1=torch.rand(1,2048,3,generator=g); x2=torch.rand(1,2048,3,generator=g)
stats(x1,x2,label='uniform vs uniform')
# trained-like: pred = gt + noise
stats(gt+0.03*torch.randn(1,2048,3,generator=g), gt, label='pred = gt + 0.03 noise')
