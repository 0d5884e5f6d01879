"""CPU replay of the auction on the C5 step clouds (bench.synth_inputs + the oracle sampler); DESIGN.md 4.4 quotes its output
(profiles/r04b_emd_cache_sim.txt).  python tools/emd_cache_sim.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vpn_oracle as O
import bench
B,K,n=1,64,32
params, gt = bench.synth_inputs(B,K,K*n,1234,'cpu')
u = O.philox_uniforms(1234, 0, B, K, n)
a = O.sample_primitives(params, [0]*K, u)[0].numpy().astype(np.float32); c = gt[0].numpy().astype(np.float32)
N=a.shape[0]
D=np.sqrt(((c[None,:,:]-a[:,None,:])**2).sum(-1)).astype(np.float32)   # [bidder, target]
for margin in (0.0, 0.01, 0.02, 0.04, 0.08):
  for rel in (0.0, 0.25):
    assign=np.full(N,-1); inv=np.full(N,-1); price=np.zeros(N,np.float32)
    memo={}; Rc=np.full(N,-1.0); lists=[None]*N
    tot_bids=0; tot_rescan=0; tot_scan_ball=0; tot_list_evals=0; lens=[]
    for it in range(50):
        U=np.nonzero(assign==-1)[0]
        d=D[U]
        val=(3-d)-price[None,:]
        bi=val.argmax(1); r=np.arange(U.size); bv=val[r,bi]; v2=val.copy(); v2[r,bi]=-np.inf; si=v2.argmax(1); sv=v2[r,si]
        if it>0:
            for k,i in enumerate(U):
                t1,t2=memo[i]; R=3-min(val[k,t1],val[k,t2])+1e-5
                tot_bids+=1
                if R>Rc[i]:
                    Rc[i]=R*(1+rel)+margin
                    need=(d[k]+price<=Rc[i])
                    lists[i]=np.nonzero(need)[0]
                    tot_rescan+=1; tot_scan_ball+=(d[k]<=Rc[i]).sum()
                # sanity: the true top two are in the list
                assert bi[k] in lists[i] and si[k] in lists[i]
                # prune list entries that can no longer matter (d + p > Rc): optional; count evals as current length
                tot_list_evals+=len(lists[i]); lens.append(len(lists[i]))
        for k,i in enumerate(U): memo[i]=(bi[k],si[k])
        if it==49: break
        inc=(bv-sv)+np.float32(0.005)
        order=np.lexsort((U,-inc.astype(np.float64),bi)); first=np.unique(bi[order],return_index=True)[1]; win=order[first]
        wi,wt=U[win],bi[win]; prev=inv[wt]; assign[prev[prev!=-1]]=-1; inv[wt]=wi; assign[wi]=wt; price[wt]+=inc[win]
    lens=np.array(lens)
    print('margin %.2f rel %.2f: bids %d  rescans %d (%.0f%%)  ball targets per rescan %.0f  list evals per bid %.1f (max list %d, p95 %d)  | scan-ball targets per bid overall %.1f'%(
        margin,rel,tot_bids,tot_rescan,100*tot_rescan/tot_bids,tot_scan_ball/max(1,tot_rescan),tot_list_evals/tot_bids,lens.max(),np.percentile(lens,95),tot_scan_ball/tot_bids))
