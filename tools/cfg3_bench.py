"""BASELINE cfg3 / M2 side numbers: MiniLM 512x128 and flat kNN at nq in {1, 64, 1024} over 1M and 10M x 384."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from eioku_amd import _lib, embed, search, synth

_lib.init(0)
gpu = torch.device("cuda:0")
out = {}
enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
g = torch.Generator(device="cpu").manual_seed(11)
ids = torch.randint(1000, 30000, (512, 128), generator=g, dtype=torch.int32).to(gpu)
mask = torch.ones((512, 128), dtype=torch.uint8, device=gpu)
for _ in range(3):
    enc.encode_ids(ids, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    enc.encode_ids(ids, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
out["minilm_512x128_ms"] = dt * 1e3
out["minilm_segments_per_s"] = 512 / dt
out["minilm_TFLOPs"] = 512 * 128 * 22.4e6 / dt / 1e12
del enc
for n in (1_000_000, 10_000_000):
    xb = synth.normal_f32(21, n, 384, gpu, l2_normalise=True)
    ix = search.IndexFlatL2(384)
    ix.attach(xb)
    for nq in (1, 64, 1024):
        q = synth.normal_f32(22, nq, 384, gpu, l2_normalise=True)
        for _ in range(2):
            ix.search(q, 10)
        torch.cuda.synchronize()
        it = 5
        t0 = time.perf_counter()
        for _ in range(it):
            ix.search(q, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        out[f"knn_{n}_nq{nq}"] = {"ms": dt * 1e3, "qps": nq / dt, "db_GBps_per_pass": n * 384 * 4 / dt / 1e9}
    del ix, xb
    torch.cuda.empty_cache()
print(json.dumps(out))
