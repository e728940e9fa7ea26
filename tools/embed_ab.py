"""Embeddings of one build, saved for a byte comparison with another build (EIOKU_HIP_LIB selects the library)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from eioku_amd import _lib, embed

_lib.init(0)
gpu = torch.device("cuda:0")
enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
res = {}
outs = []
for B, S in ((8, 128), (3, 77), (512, 128), (5, 200)):
    g = torch.Generator(device="cpu").manual_seed(B)
    ids = torch.randint(1000, 30000, (B, S), generator=g, dtype=torch.int32)
    mask = torch.ones((B, S), dtype=torch.uint8)
    for b in range(B):
        n = S - (b * 7) % (S // 2)
        mask[b, n:] = 0
        ids[b, n:] = 0
    ids, mask = ids.to(gpu), mask.to(gpu)
    o = enc.encode_ids(ids, mask)
    for _ in range(2):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    it = 10
    t0 = time.perf_counter()
    for _ in range(it):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    res[f"{B}x{S}"] = round((time.perf_counter() - t0) / it * 1e3, 4)
    outs.append(o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o))
np.savez(sys.argv[1], *outs)
print(json.dumps(res))
