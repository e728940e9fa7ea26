"""debug: candidate statistics of the list-major IVF-PQ scan on the failing test geometry"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from eioku_amd import ivfpq
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_ivfpq_gpu import clustered

d, m, nlist, n, nq, nprobe, k = 384, 48, 300, 60000, 257, 32, 20
x = clustered(7, n, d, ncl=max(8, nlist // 2), spread=0.2)
ix = ivfpq.IndexIVFPQ(d, nlist, m)
ix.train(x[: max(nlist * 40, 3000)])
ix.add(x)
ix.nprobe = nprobe
rng = np.random.default_rng(8)
q = clustered(9, nq, d, ncl=max(8, nlist // 2), spread=0.2)
q[: nq // 3] = x[rng.integers(0, n, nq // 3)]
q[-1] = -q[-1]
for cap in (2048, 8192):
    ix.cand_cap = cap
    for sl in (slice(0, nq), slice(0, nq // 3), slice(nq // 3, nq - 1), slice(nq - 1, nq)):
        D, I = ix.search(q[sl], k)
        print(cap, sl, ix.last_stats.cpu().tolist(), flush=True)
sizes = ix._pack()[1].cpu().numpy()
print("sizes min/mean/max", sizes.min(), sizes.mean(), sizes.max(), "pmax", torch.sqrt(ix._aux[2]).cpu().numpy()[:10])
