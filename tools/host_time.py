"""Host-side enqueue time of a bench step vs its wall time: python tools/host_time.py [h w]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from eioku_amd import _lib
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 640)
sys.argv = ["bench.py", "--no-cpu-baseline", "--knn-n", "0", "--height", str(h), "--width", str(w)]
args = bench.parse_args()
_lib.init(0)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
pipe = bench.Pipeline(args, dev, 0)
for i in range(4):
    pipe.step(i)
torch.cuda.synchronize()
for name in ("all", "detect", "embed", "scene"):
    if name != "all":
        args.stages = name
        pipe = bench.Pipeline(args, dev, 0)
        for i in range(4):
            pipe.step(i)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        pipe.step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:7s} host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms/step, wall {(t2 - t0) / 20 * 1e3:.3f} ms/step", flush=True)
