import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
from eioku_amd import _lib, embed
_lib.init(0); gpu = torch.device("cuda:0")
enc = embed.MiniLMEncoder(embed.random_state(embed.MINILM_L6_V2, 11))
g = torch.Generator(device="cpu").manual_seed(11)
ids = torch.randint(1000, 30000, (512, 128), generator=g, dtype=torch.int32).to(gpu)
mask = torch.ones((512, 128), dtype=torch.uint8, device=gpu)
for _ in range(3): enc.encode_ids(ids, mask)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): enc.encode_ids(ids, mask)
torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) / 5 * 1e3)
