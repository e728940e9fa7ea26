import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conftest import GOLDEN
from oracle import bert as obert
from eioku_amd import embed
g = np.load(GOLDEN / "minilm_seed11.npz")
cfg = dict(embed.MINILM_L6_V2, vocab=int(g["vocab"]))
enc = embed.MiniLMEncoder(embed.random_state(cfg, int(g["seed"])), cfg)
out = enc.encode_ids(g["ids"], g["mask"])
print("golden: err/bar =", np.abs(out - g["out"]).max() / (1e-4 * np.abs(g["out"]).max()), "shape", g["ids"].shape)
enc.close()
cfg = dict(embed.MINILM_L6_V2, vocab=3000)
state = embed.random_state(cfg, 3)
enc = embed.MiniLMEncoder(state, cfg)
rng = np.random.default_rng(5)
for B, S in [(8, 128), (3, 33), (5, 256)]:
    ids = rng.integers(1, 3000, (B, S)).astype(np.int32); mask = np.ones((B, S), np.uint8)
    want = obert.encode(state, cfg, ids, mask) if hasattr(obert, "encode") else None
    if want is None: break
    got = enc.encode_ids(ids, mask)
    print(B, S, "err/bar =", np.abs(got - want).max() / (1e-4 * np.abs(want).max()))
