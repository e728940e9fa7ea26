"""K8 time per segment (128 tokens) against the batch size: why the ingest path batches transcript segments."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from eioku_amd import _lib, embed

_lib.init(0)
gpu = torch.device("cuda:0")
enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
out = {}
for B in (8, 16, 32, 64, 128, 512):
    g = torch.Generator(device="cpu").manual_seed(11)
    ids = torch.randint(1000, 30000, (B, 128), generator=g, dtype=torch.int32).to(gpu)
    mask = torch.ones((B, 128), dtype=torch.uint8, device=gpu)
    for _ in range(3):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    it = 20
    t0 = time.perf_counter()
    for _ in range(it):
        enc.encode_ids(ids, mask)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    out[B] = {"ms": round(dt * 1e3, 4), "us_per_segment": round(dt / B * 1e6, 2)}
print(json.dumps(out))
