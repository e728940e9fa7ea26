import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eioku_amd import search, synth, _lib
_lib.init(0); gpu=torch.device('cuda:0')
n=int(sys.argv[1]) if len(sys.argv)>1 else 2_000_000; nq=int(sys.argv[2]) if len(sys.argv)>2 else 1024
xb=synth.normal_f32(21,n,384,gpu,l2_normalise=True); q=synth.normal_f32(22,nq,384,gpu,l2_normalise=True)
ix=search.IndexFlatL2(384); ix.attach(xb)
for _ in range(3): D,I=ix.search(q,10)
torch.cuda.synchronize(); print('ok')
