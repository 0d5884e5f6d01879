"""CPU replay of the auction on the C5 step clouds (bench.synth_inputs + the oracle sampler); DESIGN.md 4.4 quotes its output
(profiles/r04b_emd_balance_sim.txt).  python tools/emd_balance_sim.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vpn_oracle as O
import bench
B,K,n=1,64,32
params, gt = bench.synth_inputs(B,K,K*n,1234,'cpu')
u = O.philox_uniforms(1234, 0, B, K, n)
a = O.sample_primitives(params, [0]*K, u)[0].numpy().astype(np.float32); c = gt[0].numpy().astype(np.float32)
N=a.shape[0]; EGX,EG=32,8
mn=c.min(0); ext=c.max(0)-mn; cw=ext/np.array([EGX,EG,EG]); 
cell=np.minimum(np.floor((c-mn)/cw).astype(int), [EGX-1,EG-1,EG-1])
# per row (cy,cz): sorted x list
rows={}
for j in range(N): rows.setdefault((cell[j,1],cell[j,2]),[]).append(c[j,0])
for k in rows: rows[k]=np.sort(np.array(rows[k]))
def row_work(x, R):
    """list of (ring, ncand) for rows of the disc around bidder x with radius R (candidates = cells of the chord)"""
    cyb=int(min(max((x[1]-mn[1])//cw[1],0),EG-1)); czb=int(min(max((x[2]-mn[2])//cw[2],0),EG-1))
    cy0=int(min(max((x[1]-R-mn[1])//cw[1],0),EG-1)); cy1=int(min(max((x[1]+R-mn[1])//cw[1],0),EG-1))
    cz0=int(min(max((x[2]-R-mn[2])//cw[2],0),EG-1)); cz1=int(min(max((x[2]+R-mn[2])//cw[2],0),EG-1))
    out=[]
    for cz in range(cz0,cz1+1):
        for cy in range(cy0,cy1+1):
            ylo=mn[1]+cy*cw[1]; zlo=mn[2]+cz*cw[2]
            dy=max(ylo-x[1], x[1]-(ylo+cw[1]), 0); dz=max(zlo-x[2], x[2]-(zlo+cw[2]),0)
            h2=R*R-dy*dy-dz*dz
            if h2<0: out.append((max(abs(cy-cyb),abs(cz-czb)),-1)); continue
            h=np.sqrt(h2)
            xs=rows.get((cy,cz),np.zeros(0))
            # chord at x-cell granularity
            x0=mn[0]+np.floor((x[0]-h-mn[0])/cw[0])*cw[0]; x1=mn[0]+(np.floor((x[0]+h-mn[0])/cw[0])+1)*cw[0]
            out.append((max(abs(cy-cyb),abs(cz-czb)), int(((xs>=x0)&(xs<x1)).sum())))
    return out
# run the auction, at chosen rounds measure
assign=np.full(N,-1); inv=np.full(N,-1); price=np.zeros(N,np.float32)
memo={}
for it in range(50):
    U=np.nonzero(assign==-1)[0]
    d=np.sqrt(((c[None,:,:]-a[U,None,:])**2).sum(-1)).astype(np.float32)
    val=(3-d)-price[None,:]
    bi=val.argmax(1); r=np.arange(U.size); bv=val[r,bi]; v2=val.copy(); v2[r,bi]=-np.inf; si=v2.argmax(1); sv=v2[r,si]
    if it in (10,20,35):
        # radius from memory: previous top-2 current values
        for G,T in ((8,8),(8,4)):
            tot_mean=[];team_max=[];team_max_ring=[]; wave=[]
            for g in range(1):
                own=[k for k,i in enumerate(U) if i%G==g]
                per_bid=[]
                for k in own:
                    i=U[k]
                    if i in memo:
                        t1,t2=memo[i]; R=3-min(val[k,t1],val[k,t2])+1e-5
                    else: R=3-sv[k]+1e-5
                    rw=row_work(a[i],R)
                    cost=lambda nc: 45+27*nc if nc>=0 else 15
                    costs=[cost(nc) for _,nc in rw]
                    lanes=[sum(costs[t::T]) for t in range(T)]
                    order=np.argsort([rg for rg,_ in rw],kind='stable'); cr=[costs[o] for o in order]
                    lanes_ring=[sum(cr[t::T]) for t in range(T)]
                    per_bid.append((np.mean(lanes),max(lanes),max(lanes_ring)))
                pb=np.array(per_bid); bw=64//T
                nw=len(pb)//bw
                wmax=[pb[w*bw:(w+1)*bw,1].max() for w in range(nw)]; wmax_ring=[pb[w*bw:(w+1)*bw,2].max() for w in range(nw)]
                wmean=[pb[w*bw:(w+1)*bw,0].mean() for w in range(nw)]
                srt=pb[np.argsort(-pb[:,0])]
                wmax_sorted=[srt[w*bw:(w+1)*bw,2].max() for w in range(nw)]
                print('round %d G=%d T=%d own %d: lane mean %.0f | team max (rr) %.0f (x%.2f) | team max (ring) %.0f (x%.2f) | wave max rr %.0f (x%.2f) ring %.0f (x%.2f) ring+sorted %.0f (x%.2f)'%(
                    it,G,T,len(own),pb[:,0].mean(),pb[:,1].mean(),pb[:,1].mean()/pb[:,0].mean(),pb[:,2].mean(),pb[:,2].mean()/pb[:,0].mean(),
                    np.mean(wmax),np.mean(wmax)/np.mean(wmean),np.mean(wmax_ring),np.mean(wmax_ring)/np.mean(wmean),np.mean(wmax_sorted),np.mean(wmax_sorted)/np.mean(wmean)))
    for k,i in enumerate(U): memo[i]=(bi[k],si[k])
    if it==49: break
    inc=(bv-sv)+np.float32(0.005)
    order=np.lexsort((U,-inc.astype(np.float64),bi)); first=np.unique(bi[order],return_index=True)[1]; win=order[first]
    wi,wt=U[win],bi[win]; prev=inv[wt]; assign[prev[prev!=-1]]=-1; inv[wt]=wi; assign[wi]=wt; price[wt]+=inc[win]
