import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eioku_amd import detect as D, weights as W, _lib
from oracle import yolo as oy
_lib.init(0); gpu=torch.device('cuda:0')
for variant,nc,n,h,w in [("n",80,2,96,160),("n",1,1,64,96),("s",80,1,64,64),("m",80,1,64,96),("n",80,1,384,640)]:
    state=W.random_state(variant,nc,seed=7); det=D.Yolov8Detector(variant,nc,state)
    rng=np.random.default_rng(3); x=np.zeros((n,h,w,8),np.float16); x[...,:3]=rng.random((n,h,w,3)).astype(np.float16)
    box,cls=det.forward_raw(torch.from_numpy(x).to(gpu))
    net=oy.Net(state,*W.YOLO_VARIANTS[variant],nc)
    rb,rc=net.forward(torch.from_numpy(x[...,:3].astype(np.float32)).permute(0,3,1,2))
    for tag,G,R in (("box",box,rb),("cls",cls,rc)):
        for l,(g,r) in enumerate(zip(G,R)):
            g=g.cpu().numpy(); e=np.abs(g-r)
            print(variant,nc,tag,l,"max|r|=%.2f rms=%.3f maxerr=%.4f meanerr=%.5f  maxerr/rms=%.4f meanerr/rms=%.5f"%(np.abs(r).max(),np.sqrt((r**2).mean()),e.max(),e.mean(),e.max()/np.sqrt((r**2).mean()),e.mean()/np.sqrt((r**2).mean())))
