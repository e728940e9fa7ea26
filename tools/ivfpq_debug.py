"""debug: which query collects the most candidates in the list-major IVF-PQ filter, and why (10 M rows)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from eioku_amd import _lib, ivfpq, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq, nlist, m, nprobe, k, d = 1024, 4096, 48, 32, 10, 384
_lib.init(0)
gpu = torch.device("cuda:0")
ncl, sigma = 20000, 0.02
centres = synth.normal_f32(5, ncl, d, gpu, l2_normalise=True)
assign = torch.randint(0, ncl, (n,), device=gpu, generator=torch.Generator(device=gpu).manual_seed(6))
xb = torch.empty((n, d), dtype=torch.float32, device=gpu)
step = 2_500_000
for lo in range(0, n, step):
    hi = min(n, lo + step)
    xb[lo:hi] = centres[assign[lo:hi]] + sigma * synth.normal_f32(100 + lo // step, hi - lo, d, gpu)
qa = torch.randint(0, n, (nq,), device=gpu, generator=torch.Generator(device=gpu).manual_seed(7))
q = xb[qa] + 0.1 * sigma * synth.normal_f32(9, nq, d, gpu)
ix = ivfpq.IndexIVFPQ(d, nlist, m, device=gpu)
ix.train(xb)
for lo in range(0, n, step):
    ix.add(xb[lo:min(n, lo + step)])
ix.nprobe = nprobe
D, I = ix.search(q, k)
st = ix.last_stats.cpu().tolist()
print("stats", st)
qi = st[4]
offsets, sizes, list_codes, list_ids = ix._pack()
_, probes = ix._quantizer.search_many(q[qi:qi + 1], nprobe)
probes = probes[0].cpu().tolist()
szs = sizes.cpu().tolist()
print("query", qi, "planted row", int(qa[qi]), "final D", D[qi].cpu().tolist())
print("probe sizes", [szs[l] for l in probes])
# exact ADC distance of every code in the probed lists (decode with the codebook)
pq = ix.pq  # (m, 256, 8)
allv = []
for r, l in enumerate(probes):
    o, s = int(offsets[l]), szs[l]
    codes = list_codes[o:o + s].long()                                  # (s, m)
    dec = pq[torch.arange(m, device=gpu)[None, :], codes].reshape(s, d)  # (s, d)
    dist = ((q[qi][None, :] - ix.coarse[l][None, :] - dec) ** 2).sum(1)
    ids = list_ids[o:o + s]
    where = (ids == qa[qi]).nonzero()
    first = dist[:1024]
    kth = torch.sort(first).values[min(k, first.numel()) - 1].item() if first.numel() >= k else float("inf")
    print(f"probe {r} list {l} size {s} min {dist.min().item():.4f} p1% {torch.quantile(dist.float(), 0.01).item():.4f} median {dist.median().item():.4f}"
          f" kth-of-first-1024 {kth:.4f} planted_at {where.flatten().tolist()} |p|max {dec.norm(dim=1).max().item():.3f}")
    allv.append(dist)
allv = torch.cat(allv)
sv = torch.sort(allv).values
print("true kth", sv[k - 1].item(), "count below kth+0.01/0.02/0.05/0.1:", [(allv <= sv[k - 1] + e).sum().item() for e in (0.01, 0.02, 0.05, 0.1)])
