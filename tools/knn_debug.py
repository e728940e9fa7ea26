import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from eioku_amd import search, synth, _lib
_lib.init(0); gpu=torch.device('cuda:0')
n,d,nq,k=1_000_000,384,64,10
xb=synth.normal_f32(21,n,d,gpu,l2_normalise=True); q=synth.normal_f32(22,nq,d,gpu,l2_normalise=True)
ix=search.IndexFlatL2(d); ix.attach(xb); D,I=ix.search(q,k)
q64=q.double(); best_d=torch.full((nq,k),float('inf'),dtype=torch.float64,device=gpu); best_i=torch.full((nq,k),-1,dtype=torch.int64,device=gpu)
for lo in range(0,n,250_000):
    blk=xb[lo:lo+250_000].double(); dd=(q64*q64).sum(1)[:,None]+(blk*blk).sum(1)[None,:]-2*q64@blk.T
    cd,ci=torch.topk(dd,k,dim=1,largest=False); alld=torch.cat([best_d,cd],1); alli=torch.cat([best_i,ci+lo],1)
    o=torch.argsort(alld,dim=1,stable=True)[:,:k]; best_d,best_i=torch.gather(alld,1,o),torch.gather(alli,1,o)
err=(D.double()-best_d).abs(); rel=err/best_d.clamp(min=1e-9)
print('max abs',float(err.max()),'max rel',float(rel.max()),'id agree',float((I==best_i).float().mean()))
bad=(I!=best_i).nonzero()
print(bad[:10].tolist())
for qq,r in bad[:5].tolist(): print(qq,r,int(I[qq,r]),int(best_i[qq,r]),float(D[qq,r]),float(best_d[qq,r]), float(((q64[qq]-xb[I[qq,r]].double())**2).sum()))
print('norm check', float((xb[:5].double()**2).sum(1).max()))
import time
for nqq in (1,32,64,256,1024):
    qq=synth.normal_f32(23,nqq,d,gpu,l2_normalise=True); ix.search(qq,k); torch.cuda.synchronize(); t=time.time()
    for _ in range(5): ix.search(qq,k)
    torch.cuda.synchronize(); dt=(time.time()-t)/5; print(nqq,'ms',dt*1e3,'QPS',nqq/dt,'GB/s',n*d*4*((nqq+31)//32)/dt/1e9,'TFLOP/s',2*nqq*n*d/dt/1e12)
