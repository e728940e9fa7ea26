"""One encoder batch (B x 128 tokens), a few repetitions: the target of rocprofv3 runs on the K8 kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from eioku_amd import _lib, embed

_lib.init(0)
gpu = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
g = torch.Generator(device="cpu").manual_seed(11)
ids = torch.randint(1000, 30000, (B, 128), generator=g, dtype=torch.int32).to(gpu)
mask = torch.ones((B, 128), dtype=torch.uint8, device=gpu)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    enc.encode_ids(ids, mask)
torch.cuda.synchronize()
